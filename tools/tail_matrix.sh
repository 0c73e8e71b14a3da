# A/B of the window pipeline's knobs on the bench line (ms/window, picks/s)
cd /root/repo
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { echo "== $*"; env $E timeout 120 python bench.py --steps 320 --warmup 32 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
E="A=1" run --tail-batch 8
E="GENIE_MAIN_PRIO=-1" run --tail-batch 8
E="GENIE_MAIN_PRIO=-1" run --tail-batch 1
E="GENIE_SIDE_PRIO=1" run --tail-batch 8
E="GENIE_MAIN_PRIO=-1 GENIE_SIDE_PRIO=1" run --tail-batch 8
E="GENIE_MAIN_PRIO=0 GENIE_SIDE_PRIO=-1" run --tail-batch 8
