# needs a tuning build (-DGENIE_TUNING=1 at genie_amd/lib/libgenie_tune.so): the product library reads no environment variable
for seg in 1 4 16 64; do
  GENIE_LIB_PATH=$PWD/genie_amd/lib/libgenie_tune.so GENIE_SEG=$seg timeout 200 python bench.py --mode train --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SEG', $seg, d['ms_per_step'], d['roofline']['phase_ms'], d['four_output_step']['ms_per_step'])"
done
