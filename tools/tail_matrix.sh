# A/B of the window pipeline's knobs on the bench line (ms/window, picks/s)
cd /root/repo
run() { echo "== $E $*"; env $E timeout 120 python bench.py --steps 320 --warmup 32 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
E="A=1" run
E="GENIE_LIB_PATH=/root/repo/genie_amd/lib/libgenie_old.so" run
E="A=1" run
E="GENIE_LIB_PATH=/root/repo/genie_amd/lib/libgenie_old.so" run
E="A=1" run --mode stream
E="GENIE_LIB_PATH=/root/repo/genie_amd/lib/libgenie_old.so" run --mode stream
