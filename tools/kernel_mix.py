"""Static instruction mix of the kernels of one object file (build/<name>.o): tools/kernel_mix.py igemm [name-filter]
For straight-line kernels the static VALU : MFMA ratio is the dynamic one; 4 clocks per VALU wave-instruction, 16 per 16x16x32 MFMA."""
import collections
import re
import subprocess
import sys

obj = '/root/repo/open-solution-mapping-challenge_amd/build/%s.o' % sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
B = '/opt/rocm/lib/llvm/bin/'
subprocess.run([B + 'llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, '/tmp/mix.fatbin'], check=True)
subprocess.run([B + 'clang-offload-bundler', '--unbundle', '--type=o', '--input=/tmp/mix.fatbin', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=/tmp/mix.co'], check=True)
asm = subprocess.run([B + 'llvm-objdump', '-d', '/tmp/mix.co'], capture_output=True, text=True).stdout.split('\n')
heads = [i for i, l in enumerate(asm) if l.endswith('>:')] + [len(asm)]


def cls(l):
    m = re.match(r'\s*(\S+)', l)
    if not m:
        return None
    op = m.group(1)
    for pre, c in (('v_mfma', 'mfma'), ('v_', 'valu'), ('s_waitcnt', 'wait'), ('s_barrier', 'bar'), ('s_', 'salu'), ('ds_', 'lds'), ('global_', 'vmem'), ('buffer_', 'vmem'), ('scratch_', 'scratch')):
        if op.startswith(pre):
            return c
    return None


for a, b in zip(heads[:-1], heads[1:]):
    sym = re.search(r'<(.*)>:', asm[a]).group(1)
    name = subprocess.run(['c++filt', sym], capture_output=True, text=True).stdout.strip()
    if flt not in name:
        continue
    c = collections.Counter(cls(l) for l in asm[a + 1:b])
    body = asm[a + 1:b]
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    pre = collections.Counter(cls(l) for l in body[:mf[0]]) if mf else c
    post = collections.Counter(cls(l) for l in body[mf[-1]:]) if mf else {}
    short = re.sub(r'\(anonymous namespace\)::|msc_conv::', '', name)[:96]
    print('%-96s valu %5d (pre %4d post %4d) salu %4d mfma %4d lds %4d vmem %4d scratch %3d' % (
        short, c['valu'], pre['valu'], post.get('valu', 0), c['salu'], c['mfma'], c['lds'], c['vmem'], c['scratch']))
