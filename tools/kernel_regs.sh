#!/bin/bash
# VGPR / spill / LDS figures of the kernels of one object file: tools/kernel_regs.sh igemm [name-filter]
set -e
obj=/root/repo/open-solution-mapping-challenge_amd/build/$1.o
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$obj" /tmp/$1.fatbin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/$1.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/$1.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/$1.co | python3 -c "
import sys,re,subprocess
out=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for b in out.split('- .agpr_count:'):
    m=re.search(r'\.name:\s+(\S+)',b)
    if not m: continue
    name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
    if flt not in name: continue
    g=lambda k: re.search(k+r':\s+(\d+)',b).group(1)
    print(name[:110],'| vgpr',g(r'\.vgpr_count'),'agpr',b.split()[0],'spill',g(r'\.vgpr_spill_count'),'lds',g(r'\.group_segment_fixed_size'))
" "$2"
