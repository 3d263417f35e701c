#!/usr/bin/env python3
"""k_preprocess issues its gray loads from inline asm and waits for them two barrier intervals later with an operand-free
`s_waitcnt vmcnt(0)` (csrc/preprocess.hip, P0).  The compiler does not know that the four destination registers are
pending until that wait, so it must not read, copy or spill them in between.  This script compiles the kernel to ISA
and checks exactly that; tests/test_abi.py runs it, so a compiler that starts to touch them fails the CPU test-suite
instead of corrupting frames."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'cylinder-pose-estimation_amd', 'csrc', 'preprocess.hip')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-fast-math', '-fvisibility=hidden']


def check(hipcc='/opt/rocm/bin/hipcc'):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'pre.s')
        subprocess.check_call([hipcc] + FLAGS + ['-S', '--cuda-device-only', SRC, '-o', out], stderr=subprocess.DEVNULL)
        lines = open(out).read().split('\n')
    loads = [i for i, l in enumerate(lines) if re.match(r'\s*global_load_(ubyte|dword) v\d+, v\[\d+:\d+\], off', l)]
    waits = [i for i, l in enumerate(lines) if l.strip() == 's_waitcnt vmcnt(0)' and ';;#ASMSTART' in lines[i - 1]]
    if len(loads) != 5 or len(waits) != 1 or waits[0] < loads[-1]:
        return f'unexpected shape: {len(loads)} asm loads, {len(waits)} asm waits'
    regs = sorted({re.search(r'global_load_\w+ (v\d+),', lines[i]).group(1) for i in loads})
    if len(regs) != 4:
        return f'expected 4 destination registers, found {regs}'
    pat = re.compile(r'\b(' + '|'.join(regs) + r')\b')
    spans = re.compile(r'v\[(\d+):(\d+)\]')
    nums = {int(r[1:]) for r in regs}
    for i in range(loads[-1] + 1, waits[0]):
        l = lines[i].split(';')[0]
        if pat.search(l) or any(nums & set(range(int(a), int(b) + 1)) for a, b in spans.findall(l)):
            return f'line {i + 1} touches a pending load register before the wait: {lines[i].strip()}'
    if any('scratch_' in l for l in lines):
        return 'the kernel spills to scratch'
    return None


if __name__ == '__main__':
    err = check()
    print('ok' if err is None else 'FAIL: ' + err)
    sys.exit(0 if err is None else 1)
