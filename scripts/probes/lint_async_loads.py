"""Lint for kernels that request registers with inline-asm loads and wait for them later with a counted `s_waitcnt vmcnt(N)`.

hipcc does not know that the destination registers of such a load are pending: between the request and the wait it may legally copy, spill or
split their live range — and would read them before the data has arrived (seen in conv1x1_bwd.hip, round 4: `v_mov_b64` copies in front of a wait).
This script compiles a source file to gfx950 assembly and walks every kernel's control-flow graph: from each inline-asm `global_load_dwordx4 vD, ...`
it follows all paths until an inline-asm `s_waitcnt vmcnt(..)` and reports any instruction on the way that reads (or overwrites) a register of vD,
other than the address operand of the load itself.  Exit status 1 on a finding.

Usage: python scripts/lint_async_loads.py hd_yolo_amd/csrc/conv1x1_bwd.hip [more.hip ...]
(no GPU needed; the CPU test suite runs it: tests/test_build.py)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'hd_yolo_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-I' + CSRC, '-I' + os.path.join(ROOT, 'include')]

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse_kernels(asm):
    """-> {name: [(label | None, text, in_asm)]} instruction list per kernel"""
    kernels, cur, name, in_asm = {}, None, None, False
    for line in asm.split('\n'):
        t = line.split(';')[0].rstrip() if ';;#' not in line else line.strip()
        s = t.strip()
        if re.match(r'^_Z\w+:$', s):
            name, cur, in_asm = s[:-1], [], False
            kernels[name] = cur
            continue
        if cur is None or not s:
            continue
        if s.startswith('.Lfunc_end'):
            cur = None
            continue
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if re.match(r'^\.LBB\w+:$', s):
            cur.append((s[:-1], None, False))
            continue
        if s.startswith('.') or s.startswith(';'):
            continue
        cur.append((None, s, in_asm))
    return kernels


def lint_kernel(name, ins):
    labels = {lab: i for i, (lab, _, _) in enumerate(ins) if lab}
    findings = []

    def succ(i):
        _, t, _ = ins[i]
        if t is None:
            return [i + 1]
        op = t.split()[0]
        if op == 's_endpgm':
            return []
        if op == 's_branch':
            return [labels[t.split()[1]]]
        if op.startswith('s_cbranch'):
            return [labels[t.split()[1]], i + 1]
        return [i + 1]

    for i, (_, t, in_asm) in enumerate(ins):
        if not (in_asm and t and t.startswith('global_load_dword')):
            continue
        ops = t.split(None, 1)[1].split(',')
        dst = regs_of(ops[0])
        seen, stack = set(), [j for j in succ(i)]
        while stack:
            j = stack.pop()
            if j in seen or j >= len(ins):
                continue
            seen.add(j)
            _, u, u_asm = ins[j]
            if u is not None:
                if u_asm and u.startswith('s_waitcnt') and 'vmcnt' in u:
                    continue                                   # this path is settled
                if u_asm and u.startswith('global_load_dword'):
                    touched = regs_of(u.split(None, 1)[1].split(',', 1)[1]) & dst      # another request may only not READ them
                else:
                    touched = regs_of(u.split(None, 1)[1] if ' ' in u else '') & dst
                if touched:
                    findings.append(f'{name}: `{u}` touches v{sorted(touched)} requested by `{t}` before any counted wait')
                    continue
            stack.extend(succ(j))
    return findings


def lint_file(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        subprocess.check_call([hipcc] + FLAGS + ['-S', '--cuda-device-only', src, '-o', out], stderr=subprocess.DEVNULL)
        asm = open(out).read()
    findings, nreq = [], 0
    for name, ins in parse_kernels(asm).items():
        nreq += sum(1 for _, t, a in ins if a and t and t.startswith('global_load_dword'))
        findings += lint_kernel(name, ins)
    return nreq, findings


if __name__ == '__main__':
    bad = 0
    for src in sys.argv[1:]:
        nreq, findings = lint_file(src)
        print(f'{src}: {nreq} inline-asm register requests, {len(findings)} findings')
        for f in findings:
            print('  ' + f)
        bad += len(findings)
    sys.exit(1 if bad else 0)
