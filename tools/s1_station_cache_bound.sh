# Upper bound of what ANY cache of the station neighbours' hidden states can give k_stage1_h2, measured with
# the tuning build's ablation bits: 8192 = the eight station-neighbour units cost nothing (no row load, no MFMA, no accumulate; wrong results);
# 8192 + 16384 = ... and are replaced by what an LDS cache would cost per unit instead (address, four ds_read_b128 of a 136-B-pitch row, sixteen adds).
# Usage (GPU box): bash tools/s1_station_cache_bound.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GENIE_LIB_PATH=$PWD/genie_amd/lib/libgenie_tune.so
for v in 0 8192 24576 0 8192 24576; do GENIE_ABLATE=$v python tools/s1_time.py cfg2_200x10k 80 2>&1 | tail -1; done
