cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfs
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pfs -o d -- python3 $R/bench.py --forward-sweep --sweep-batches ${1:-8192} > /dev/null 2>&1
DB=$(ls /tmp/pfs/*/*.db /tmp/pfs/*.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_summary.py $DB > $R/gpurun_out/fs${1:-8192}_summary.md 2>&1
