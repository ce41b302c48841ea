#!/bin/bash
# where does the lane-per-player bitboard kernel (with its replay epilogue) start to beat byte slabs on 40x40 and 20x20?
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for T in 16 32 64 128; do timeout -k 10 300 python3 tools/kernel_ab.py 40 $T || exit 1; done
for T in 64 256 1024; do timeout -k 10 300 python3 tools/kernel_ab.py 20 $T || exit 1; done
