"""ISA check of the kernels that read LDS through inline asm with deferred waits (transpose reads, ds_read_b128 under
counted lgkmcnt): compiles every .hip of gridmm_amd/csrc to gfx950 assembly (device only, a few seconds per file) and walks
each kernel in program order with the LDS queue the hardware keeps (in-order returns: `s_waitcnt lgkmcnt(n)` retires all
but the youngest n LDS operations).  Any vector instruction that reads the destination of an LDS read still outstanding
is a hazard: the compiler treats an asm-issued read as complete and may COPY its registers before the asm wait that
follows (this is what made the single-group accumulator form of aggregate_pipe.hip return run-dependent sums, round 3).
usage: python tools/check_lds_hazards.py [file.hip ...]      exit code 1 on a hazard"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "gridmm_amd", "csrc")


def _regs(tok):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def hazards(asm_text):
    found = []
    for km in re.finditer(r"^(_Z\w*):[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.M | re.S):
        name, queue = km.group(1), []            # queue: destination registers of outstanding LDS operations, in issue order
        for line in km.group(2).split("\n"):
            t = line.strip()
            if re.match(r"^\.?L?BB\d+_\d+:", t):
                queue = []                         # block entry: the compiler's own waits cover what it knows about
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            op = t.split()[0]
            args = t[len(op):].split(",")
            if op.startswith("ds_"):
                is_read = op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_swizzle", "ds_permute"))
                queue.append(_regs(args[0]) if is_read else set())
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    queue = queue[len(queue) - n:] if n else []
                continue
            if op.startswith("s_"):
                continue
            srcs = set()
            for a in args[1:]:
                srcs |= _regs(a)
            if any(srcs & d for d in queue):
                found.append((name, t))
    return found


def check_file(path, flags=("-O3", "-std=c++17")):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *flags, "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, path]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s" % (path, r.stderr[-2000:]))
        return hazards(open(out).read())


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = 0
    for f in files:
        h = check_file(f)
        print("%-28s %s" % (os.path.basename(f), "ok" if not h else "%d hazard(s)" % len(h)))
        for name, t in h[:6]:
            print("    %s : %s" % (name[:80], t))
        bad += len(h)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
