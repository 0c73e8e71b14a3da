# A/B of the window pipeline's knobs on the bench line (ms/window, picks/s)
cd /root/repo
run() { echo "== $*"; env "$@" timeout 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run GENIE_TAILS=1
run GENIE_TAILS=2
run GENIE_TAILS=3
run GENIE_TAILS=2 GENIE_SIDE_PRIO=-1
run GENIE_TAILS=3 GENIE_SIDE_PRIO=-1
run GENIE_TAILS=2 GENIE_BPC2=16
