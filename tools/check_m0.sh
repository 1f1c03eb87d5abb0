#!/bin/bash
# The inflate symbol loop (inflate_wave.h, iw_run) keeps the output position in m0 and cannot name m0 as clobbered (a reserved
# register).  This lists every instruction of the compiled device code that touches m0: all of them must come from that loop.
set -e
D=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o $D/bgzf.s "$(dirname "$0")/../strling_amd/csrc/bgzf.hip" 2>/dev/null
grep "m0" $D/bgzf.s | sed 's/^\s*//' | awk '{print $1}' | sort | uniq -c
n=$(grep "m0" $D/bgzf.s | grep -vc "v_writelane_b32\|s_sub_u32 s95, m0\|s_mov_b32 s[0-9]*, m0\|s_and_b32 s92, m0, 63\|s_add_u32 m0, m0\|v_add_u32_e32 v[0-9]*, m0\|s_mov_b32 m0, s[0-9]*\|v_sub_u32_e32 v[0-9]*, m0\|v_cmp_lt_u32_e32 vcc, m0\|s_cmp_\w* .*m0\|s_add_u32 s93, m0" || true)
echo "instructions touching m0 outside the symbol loop's forms: $n"
[ "$n" = "0" ]
