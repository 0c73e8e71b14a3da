# Production code generation, split against one-wave pass 1' (round 6). needs: python -c "from genie_amd import _lib; _lib.build(extra_flags=['-DGENIE_B1_SPLIT_DEFAULT=0'], out_path='genie_amd/lib/variants/libgenie_b1_one_wave.so')"
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in $R/genie_amd/lib/libgenie_hip.so $R/genie_amd/lib/variants/libgenie_b1_one_wave.so; do
  echo "== $(basename $lib) run $rep"
  GENIE_LIB_PATH=$lib timeout 300 python $R/bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['phase_ms'], d['four_output_step']['ms_per_step'])"
done; done
