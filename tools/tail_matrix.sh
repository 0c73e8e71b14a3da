# Same-box A/B harness for the bench line (ms/window, picks/s): box-to-box spread is +-1.5 %, so every comparison in DESIGN.md
# section 5 alternates the two variants inside ONE gpurun call. Edit the `E=... run ...` lines: environment knobs and / or
# GENIE_LIB_PATH=<another build of the library> (e.g. `git stash; python __graft_entry__.py build; cp genie_amd/lib/libgenie_hip.so
# genie_amd/lib/libgenie_old.so; git stash pop; python __graft_entry__.py build`).
cd /root/repo
run() { echo "== $E $*"; env $E timeout 120 python bench.py --steps 320 --warmup 32 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
E="A=1" run
E="GENIE_LIB_PATH=/root/repo/genie_amd/lib/libgenie_old.so" run
E="A=1" run
E="GENIE_LIB_PATH=/root/repo/genie_amd/lib/libgenie_old.so" run
E="A=1" run --mode stream
E="GENIE_LIB_PATH=/root/repo/genie_amd/lib/libgenie_old.so" run --mode stream
