for wg in 2 3 4; do
  GENIE_TRAIN_WG=$wg GENIE_LIB_PATH=$PWD/genie_amd/lib/libgenie_tune.so timeout 200 python bench.py --mode train --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WG', $wg, d['ms_per_step'], d['roofline']['phase_ms'], d['four_output_step']['ms_per_step'])"
done
