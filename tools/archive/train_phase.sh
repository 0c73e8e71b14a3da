# phase times of the training step: bash tools/train_phase.sh [extra bench args]
timeout 300 python bench.py --mode train --steps ${STEPS:-30} --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['roofline']['phase_ms'], 'four_output', d['four_output_step']['ms_per_step'])"
