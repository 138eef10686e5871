#!/bin/bash
# Device ISA of the library (extra -D flags as arguments) and the register / scratch use of the kernels matching $KPAT.
#   KPAT=k_spec tools/isa_regs.sh -DKSPEC_FUSE_K1
set -e
cd "$(dirname "$0")/../ksfd_amd/csrc"
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -S --cuda-device-only -o /tmp/isa/ksfd_x.s ksfd_hip.hip 2>&1 | grep -E "error" -A5 || true
python3 - <<'PY'
import re, os
pat = os.environ.get('KPAT', 'k_spec')
txt = open('/tmp/isa/ksfd_x.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S*%s\S*)' % pat, txt):
    blk = txt[m.start():m.start() + 4000]
    print('%-60s vgpr %s scratch %s' % (m.group(1)[:60], re.findall(r'\.amdhsa_next_free_vgpr (\d+)', blk)[0], re.findall(r'\.amdhsa_private_segment_fixed_size (\d+)', blk)[0]))
PY
