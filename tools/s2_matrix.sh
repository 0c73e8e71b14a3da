# Stage-2 kernel time under the scheduling switches (round-3 experiment, profiles/r03_k_s2mat.txt): bash tools/s2_matrix.sh
# (The GENIE_SEG2 rows of that profile came from an experiment-only switch of commit f1fa2e4.)
cd /root/repo
python tools/s2_time.py cfg2_200x10k 40 2>/dev/null | tail -1
GENIE_S2_WGMAP=1 python tools/s2_time.py cfg2_200x10k 40 2>/dev/null | tail -1
for b in 2 4; do GENIE_BPC2=$b python tools/s2_time.py cfg2_200x10k 40 2>/dev/null | tail -1; done
