"""ISA lint for a compiler hazard met in this round (AMD clang 22, gfx950): the register allocator may move a VGPR whose
value is live in ALL lanes into an AGPR (v_accvgpr_write) inside a divergent region, i.e. while EXEC is narrowed by
s_and_saveexec / s_mov exec.  Only the active lanes are saved; when the VGPR is then reused under full EXEC and the value
is read back after the region, the inactive lanes come back as garbage.  Seen in a profiling build of the f64 fill
kernel: a zero offset register saved with lane 63 masked off -> wild global address on the next worklist pop -> GPU
memory fault.  (The shipped kernels only show the benign forms: a register copied onto itself, or a temporary that is
consumed inside the same region.)

    python tools/lint_exec_spills.py file.s [kernel-name-substring]      exit status 1 = harmful pattern found

Per kernel, in program order: a stack of the SGPR pairs holding a saved EXEC (`s_*_saveexec_b64 sX` and
`s_mov_b64 sX, exec` push; `s_or_b64 exec, exec, sX` / `s_mov_b64 exec, sX` pop down to sX).  A v_accvgpr_write aN
issued while the stack is non-empty is reported when, after its region has been left, aN is READ before it is written
again -- unless the write stores back what was just read from the same aN.  Linear scan (ignores the CFG): a heuristic,
good enough to catch the pattern above; run by tests/test_build_lint.py on every kernel of the library.

The same hazard through SCRATCH (a "Folded Spill" scratch_store under narrowed EXEC, reloaded after the region): a store
under narrowed EXEC is harmless when the slot already holds a value of every lane (an earlier store under full EXEC: that
is how a per-lane PHI of a spilled value looks, e.g. the 100-byte private segment of fill_round_kernel<float,false,1>
at commit 2c30b20); it is reported when NO full-EXEC store precedes it and the slot is reloaded after the region.
`private_segments(path)` lists every kernel's .private_segment_fixed_size so the tests can also require "no scratch at
all" of the shipped fill kernels."""
import re
import sys

SAVE = re.compile(r"s_(?:and|andn2|or|xor)_saveexec_b64 (s\[\d+:\d+\])")
COPY = re.compile(r"s_mov_b64 (s\[\d+:\d+\]), exec")
REST = re.compile(r"(?:s_or_b64 exec, exec, |s_mov_b64 exec, )(s\[\d+:\d+\])")
WR = re.compile(r"v_accvgpr_write_b32 (a\d+), (v\d+)")
RD = re.compile(r"v_accvgpr_read_b32 (v\d+), (a\d+)")
USE = re.compile(r"\ba\[?(\d+)(?::(\d+))?\]?")
SST = re.compile(r"scratch_store_dword(x\d)? off, v\[?\d+(?::\d+)?\]?, off(?: offset:(\d+))?")
SLD = re.compile(r"scratch_load_dword(x\d)? v\[?\d+(?::\d+)?\]?, off, off(?: offset:(\d+))?")


def _slots(m):
    n = int(m.group(1)[1:]) if m.group(1) else 1
    off = int(m.group(2) or 0)
    return [off + 4 * k for k in range(n)]


def kernels(path):
    name, body = None, []
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", s)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append((ln, s))
        if s.startswith("s_endpgm"):
            yield name, body
            name = None


def lint_kernel(body):
    stack, depth_at, writes = [], [], []
    for i, (ln, s) in enumerate(body):
        m = SAVE.match(s)
        if m:
            stack.append(m.group(1))
        m = COPY.match(s)
        if m:
            stack.append(m.group(1))
        m = REST.match(s)
        if m and m.group(1) in stack:
            del stack[stack.index(m.group(1)):]
        depth_at.append(len(stack))
        m = WR.match(s)
        if m and stack:
            prev = " ".join(t for _, t in body[max(0, i - 4):i])
            if re.search(r"v_accvgpr_read_b32 %s, %s\b" % (m.group(2), m.group(1)), prev):
                continue   # stores back what it just read from the same AGPR
            writes.append((i, ln, m.group(1), len(stack), s))
    bad = []
    # scratch spill slots: stores under narrowed EXEC into a slot no full-EXEC store has filled, reloaded after the region
    full, partial = set(), {}
    for i, (ln, s) in enumerate(body):
        m = SST.match(s)
        if m:
            for slot in _slots(m):
                if depth_at[i] == 0:
                    full.add(slot)
                    partial.pop(slot, None)
                elif slot not in full and slot not in partial:
                    partial[slot] = (ln, depth_at[i], s)
            continue
        m = SLD.match(s)
        if m:
            for slot in _slots(m):
                if slot in partial and depth_at[i] < partial[slot][1]:
                    pl, pd, ps = partial.pop(slot)
                    bad.append((pl, pd, ps, ln, s))
    for i, ln, areg, depth, s in writes:
        n = int(areg[1:])
        left = False
        for j in range(i + 1, len(body)):
            t = body[j][1]
            if depth_at[j] < depth:
                left = True
            w = WR.match(t)
            if w and w.group(1) == areg:
                break                      # overwritten before anybody outside read it
            if left:
                reads = False
                r = RD.match(t)
                if r and r.group(2) == areg:
                    reads = True
                elif not t.startswith("v_accvgpr_write"):
                    for u in USE.finditer(t.split(";")[0]):
                        lo = int(u.group(1))
                        hi = int(u.group(2)) if u.group(2) else lo
                        if lo <= n <= hi:
                            reads = True
                if reads:
                    bad.append((ln, depth, s, body[j][0], t))
                    break
    return bad


def lint(path, only=None):
    out = []
    for name, body in kernels(path):
        if only and only not in name:
            continue
        for ln, depth, s, ln2, t in lint_kernel(body):
            out.append((name, ln, depth, s, ln2, t))
    return out


def private_segments(path):
    """{kernel name: .private_segment_fixed_size} from the kernel descriptors' metadata."""
    out, size = {}, None
    for line in open(path):
        m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
        if m:
            size = int(m.group(1))
        m = re.match(r"\s+(?:- )?\.name:\s+(\S+)", line)
        if m and size is not None:
            pass
        m2 = re.match(r"\s+\.symbol:\s+(\S+)\.kd", line)
        if m2 and size is not None:
            out[m2.group(1)] = size
            size = None
    return out


if __name__ == "__main__":
    res = lint(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    for name, ln, depth, s, ln2, t in res:
        print("HARMFUL? %s\n   line %d (exec narrowed, depth %d): %s\n   read after the region at line %d: %s" % (name, ln, depth, s, ln2, t))
    print("%d harmful spill pattern(s)" % len(res))
    sys.exit(1 if res else 0)
