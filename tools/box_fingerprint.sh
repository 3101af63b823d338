#!/bin/bash
# Which box is this?  (gpurun hands out a fresh box per call; the mask kernel's rates differ by box: profiles/r06_mask_alloc.md)
{ echo "== $(date -u +%FT%TZ) host $(hostname) kernel $(uname -r) nproc $(nproc)"
  rocm-smi --showuniqueid --showserial --showbus 2>&1 | grep -i "GPU\[" | head -6
  rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -i "GPU\[" | head -4
  rocm-smi --showclocks 2>&1 | grep -i "GPU\[" | head -12
  rocm-smi --showtemp --showpower --showmaxpower 2>&1 | grep -i "GPU\[" | head -10
  rocm-smi --showvbios --showdriverversion 2>&1 | grep -i "GPU\[\|driver" | head -4
  rocm-smi --showmeminfo vram 2>&1 | grep -i "GPU\[" | head -3
  rocm-smi --showrasinfo all 2>&1 | grep -i "umc\|hbm" | head -6
} 2>&1
