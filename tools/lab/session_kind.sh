#!/bin/bash
# lab: what kind of session is this?  Clocks / power state as rocm-smi shows them, then the store shapes on pools of separately allocated
# 16-frame chunks (neighbours / every fourth / random / dealt) — in a "good" session every fourth takes ~7.0 TB/s, in a "slow" one nearly everything 5.4 - 5.7.
R="${GRAFT_REPO_ROOT:-$(pwd)}"
hostname; (rocm-smi --showuniqueid --showserial --showbus 2>&1 | grep "GPU\[" | head -4) || true
(rocm-smi --showclocks --showperflevel --showpower --showtemp --showmemuse --showmemvendor 2>&1 | grep -v "^=\|^$" | head -60) || true
(rocm-smi --showcomputepartition --showmemorypartition 2>&1 | grep -v "^=\|^$" | head -10) || true
(cat /sys/class/drm/card*/device/mem_info_vram_used 2>/dev/null | head -3) || true
LAB_SPREAD=1 LAB_CH=16 timeout -k 10 150 $R/tools/front_lab.bin 512 128
(rocm-smi --showclocks 2>&1 | grep -i "mclk\|fclk\|sclk\|socclk" | head -12) || true
