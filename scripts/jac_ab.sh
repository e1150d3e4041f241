#!/bin/bash
# A/B of the J-assembly kernel's experiment switches on ONE box (box-to-box variation is larger
# than most of the effects): MMX_JAC_NT (streaming stores), MMX_JAC_ZERO_PHASE (0 first, 1 alternating, 2 last)
for nt in 0 1; do for zp in 0 1 2; do
  echo "== MMX_JAC_NT=$nt MMX_JAC_ZERO_PHASE=$zp"
  MMX_JAC_NT=$nt MMX_JAC_ZERO_PHASE=$zp python scripts/jac_time.py ${@:-4096 8192 32768}
done; done
