R=$(pwd); O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_fetch /tmp/pmc_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -o p -- python3 $R/bench.py --kernels-only > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -o p -- python3 $R/bench.py --kernels-only > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write > $O/pmc_per_launch.json
cp $O/pmc_per_launch.json $R/profiles/pmc_per_launch.json
cd $R && timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; cut -c1-160 $O/bench.json
