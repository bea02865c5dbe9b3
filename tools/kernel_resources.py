"""Registers, LDS and scratch of every kernel in the built objects (stable-diffusion_amd/build/*.o): the code-object metadata
hipcc wrote, no GPU needed.  `python tools/kernel_resources.py [pattern]` -- run after build.py.

A gfx950 SIMD has 512 VGPRs per lane (arch + accumulation registers, allocated together in steps of 8): waves per SIMD =
min(8, 512 // vgpr_count).  A helper kernel that is a latency chain (LayerNorm, GroupNorm apply, split-K reduce) pays for a
low number directly: round 2 found LayerNorm at 108 VGPRs for rows that need 28 (profiles/epilogue_16byte_r02.txt (5)),
and a "prefetch" that took it to 380."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def kernels_of(obj, tmp):
    fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
    r = subprocess.run([f'{LLVM}/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fat], capture_output=True)
    if r.returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return []
    r = subprocess.run([f'{LLVM}/clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        f'--input={fat}', f'--output={co}', '--unbundle'], capture_output=True)
    if r.returncode:
        return []
    notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
    out = []
    for k in re.split(r'\n\s+- \.agpr_count:', notes)[1:]:
        g = lambda key: int(re.search(r'\.' + key + r':\s+(\d+)', k).group(1))
        name = re.search(r'\.name:\s+(\S+)', k).group(1)
        out.append(dict(name=name, agpr=int(k.split('\n')[0].strip()), vgpr=g('vgpr_count'), sgpr=g('sgpr_count'),
                        lds=g('group_segment_fixed_size'), scratch=g('private_segment_fixed_size'),
                        spill=g('vgpr_spill_count')))
    return out


def main():
    pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(ROOT, 'stable-diffusion_amd', 'build', '*.o'))):
            for k in kernels_of(obj, tmp):
                k['obj'] = os.path.basename(obj)
                rows.append(k)
    names = subprocess.run(['c++filt'], input='\n'.join(k['name'] for k in rows), capture_output=True, text=True).stdout.split('\n')
    print(f'{"kernel":96s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"LDS":>7s} {"scratch":>7s} {"waves/SIMD":>10s} {"WG/CU by LDS":>12s}')
    for k, d in zip(rows, names):
        d = d.replace('sdmi::(anonymous namespace)::', '').replace('void ', '').replace('sdmi::', '')
        d = re.sub(r'\(.*$', '', d)
        if pat and not pat.search(d):
            continue
        waves = min(8, 512 // max(k['vgpr'], 1))
        by_lds = 'any' if k['lds'] == 0 else str(min(16, (160 * 1024) // k['lds']))
        print(f'{d[:96]:96s} {k["vgpr"]:5d} {k["agpr"]:5d} {k["sgpr"]:5d} {k["lds"]:7d} {k["scratch"]:7d} {waves:10d} {by_lds:>12s}')


if __name__ == '__main__':
    main()
