cd /root/repo
for seg in 1 2 4 8 16 32 64; do GENIE_SEG2=$seg python tools/s2_time.py cfg2_200x10k 40 2>/dev/null | tail -1; done
for seg in 1 8; do GENIE_S2_WGMAP=1 GENIE_SEG2=$seg python tools/s2_time.py cfg2_200x10k 40 2>/dev/null | tail -1; done
for b in 2 4; do GENIE_BPC2=$b python tools/s2_time.py cfg2_200x10k 40 2>/dev/null | tail -1; done
for seg in 8 16; do GENIE_BPC2=2 GENIE_SEG2=$seg python tools/s2_time.py cfg2_200x10k 40 2>/dev/null | tail -1; done
