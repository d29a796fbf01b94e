run() { echo "$1 $(timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline $2 | cut -c186-194)"; }
run base ""
for v in 224 288 320; do run "wgrad_blocks=$v" "--tune 3=$v"; done
run base ""
for v in 640 896 1024; do run "fwd_blocks=$v" "--tune 0=$v"; done
for v in 1280 1792 2048; do run "dst_blocks=$v" "--tune 11=$v"; done
run base ""
for v in 384 640 768; do run "src_blocks=$v" "--tune 12=$v"; done
for v in 128 384 512; do run "rd_blocks=$v" "--tune 13=$v"; done
run base ""
