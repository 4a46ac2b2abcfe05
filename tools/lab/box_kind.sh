#!/bin/bash
# lab: what kind of box is this?  The stores-only frame placement lab, then the all-solid workload with the frames in one pool / one allocation each.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
(rocm-smi --showbus --showserial 2>/dev/null | grep -E "PCI Bus|Serial") || true
timeout -k 10 120 tools/front_lab.bin 512 2>&1 | grep -E "tile-major|segment|frame 0" | grep -v "T  2048"
WORKLOADS=msvideo1_16_1080p_keyframes_solid ROUNDS=1 tools/lab/pool_ab.sh
