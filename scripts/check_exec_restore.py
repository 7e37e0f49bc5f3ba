#!/usr/bin/env python3
"""A release gate that needs no GPU: scan the gfx950 code objects of the build for VGPR writes that sit at the head of a reconvergence block ABOVE the instruction
that restores EXEC.

Root cause of the wrong films / faults of rounds 4 and 5 (profiles/r05_isaemu_root_cause.md): ROCm 7.2.0's clang 22, greedy VGPR allocator, placed a live-range
copy (`v_mov_b64 v[150:151], v[10:11]`) at the top of the block in which the lanes of an `if` rejoin, three scalar instructions BEFORE the
`s_or_b64 exec, exec, s[0:1]` that re-enables the lanes which skipped the branch.  Those lanes never get the copy and read a stale register later: one component
of a float3, half of a pointer, half of a return address.  The pattern is syntactic:

    <label: the target of the s_cbranch_execz that skips the `then` block>
        ... scalar instructions ...
        v_mov_b64_e32 v[150:151], v[10:11]          <-- executes under the `then` mask
        ... scalar instructions ...
        s_or_b64 exec, exec, s[0:1]                 <-- the lanes that skipped come back HERE

so every build can be checked for it: a block that starts at a branch target and reaches an EXEC-restoring instruction through straight-line code must not write
a vector register on the way (v_readlane / v_readfirstlane / v_cmp write scalar registers; v_writelane writes one named lane whatever EXEC is: the SGPR spills).

    python scripts/check_exec_restore.py [object or code object or disassembly ...]      (default: every pbrt-v2_amd/build/hpt_kernels_*.o)
exit status 1 if a definition above a restore is found."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
EXEC_RESTORE = re.compile(r"^(s_or_b64 exec, exec, |s_or_saveexec_b64 |s_andn2_saveexec_b64 |s_xor_b64 exec, exec, |s_mov_b64 exec, |s_andn2_b64 exec, exec, |s_and_b64 exec, exec, |s_xor_saveexec_b64 |s_and_saveexec_b64 )")
SCALAR_DEST_VALU = ("v_readlane_b32", "v_readfirstlane_b32", "v_cmp", "v_writelane_b32")
STOP = ("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")


# Sites of this shape the build is known to ship with: none.  (Round 5 found one — `v_mov_b64 v[118:119], v[110:111]` in the free-running configuration 0 of the basic set — and
# removed it with a barrier compiled into that instantiation only, hpt_kernels_impl.h HPT_CODEGEN_NUDGE: profiles/r05_isaemu_root_cause.md §4.)
KNOWN_SITES = set()


def new_sites(paths=None):
    """the definitions above an EXEC restore in the build's kernel units that are not in KNOWN_SITES -> set of (unit, kernel, instruction)"""
    import concurrent.futures
    paths = paths or device_objects()
    sites = set()
    with concurrent.futures.ProcessPoolExecutor(max_workers=min(8, len(paths))) as pool:       # (a unit is a few hundred thousand to 1.7 M instructions of text: one process each)
        for p, (nf, ni, found) in zip(paths, pool.map(_scan_path, paths)):
            if nf == 0 or ni == 0:
                raise RuntimeError("no code found in " + p)
            sites |= {(os.path.basename(p), f[0], f[3]) for f in found if f[-1].startswith("DEFINES")}
    return sites - KNOWN_SITES


def device_objects():
    """every object of the build that carries gfx950 code: the kernel units, the wavefront pipeline, the device BVH builder, the calibration and exchange kernels"""
    build = os.path.join(ROOT, "pbrt-v2_amd", "build")
    return sorted(glob.glob(os.path.join(build, "hpt_kernels*.o"))) + [p for p in (os.path.join(build, n) for n in ("hpt_wavefront.o", "hpt_bvh_gpu.o", "hpt_calib.o", "hpt_multi.o")) if os.path.exists(p)]


def _scan_path(p):
    d = disassembly_of(p)
    return scan(d) if d is not None else (0, 0, [])


def disassembly_of(path, tmp="/tmp/check_exec_restore"):
    if path.endswith(".s"):
        return open(path)
    os.makedirs(tmp, exist_ok=True)
    co = path
    if path.endswith(".o") or path.endswith(".so"):
        import hashlib
        tag = hashlib.sha1(os.path.abspath(path).encode()).hexdigest()[:10] + "_" + os.path.basename(path)      # (the Makefile gates objects of several builds in parallel)
        fat, co = os.path.join(tmp, tag + ".fat"), os.path.join(tmp, tag + ".co")
        subprocess.check_call([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        if os.path.getsize(fat) == 0:
            return None                                      # an object without device code (host-only translation unit)
        r = subprocess.run([BUNDLER, "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"], capture_output=True, text=True)
        if r.returncode != 0:
            if "Can't find bundles" in r.stderr:
                return None                                  # a fat binary without a gfx950 code object: nothing to scan
            raise RuntimeError(r.stderr)
    return subprocess.Popen([OBJDUMP, "-d", co], stdout=subprocess.PIPE, text=True).stdout


def writes_vgpr(text):
    op = text.split(None, 1)[0]
    if op.startswith(SCALAR_DEST_VALU):
        return False
    if op.startswith("v_"):
        return True
    if re.match(r"^(global|flat|scratch|buffer)_load|^ds_read|^ds_bpermute|^ds_permute|^(global|flat)_atomic.*\b(sc0|glc)\b", text):
        return True
    return False


def dest_vgprs(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return set()
    if parts[0].startswith("v_swap_b32"):            # D <-> S0: both registers are written
        return {int(x) for x in re.findall(r"\bv(\d+)\b", parts[1])}
    first = parts[1].split(",")[0].strip()
    m = re.match(r"^v\[(\d+):(\d+)\]$", first)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", first)
    return {int(m.group(1))} if m else set()


def check_function(name, insns):
    """insns: [(addr, text)] of one function -> [(label addr, offending addr, offending text, restore addr, restore text)]
    The structure looked for is the compiler's own `if`: s_and_saveexec_b64 s[a:b], <cond> ; s_cbranch_execz L ; <then> ; L: ... ; s_or_b64 exec, exec, s[a:b]"""
    pos = {a: i for i, (a, t) in enumerate(insns)}
    found = []
    for i, (addr, text) in enumerate(insns):
        if not text.startswith("s_cbranch_execz"):
            continue
        off = int(text.split()[-1])
        if off >= 32768:
            continue                                  # (backward: a loop, not the skip of a `then` block)
        label = addr + 4 + 4 * off
        saved = None
        for a0, t0 in reversed(insns[max(0, i - 3):i]):
            m = re.match(r"^s_(and|andn2|or|xor)_saveexec_b64 (s\[\d+:\d+\]|vcc),", t0)
            if m:
                saved = m.group(2)
                break
        j = pos.get(label)
        if saved is None or j is None:
            continue
        pending = []
        for a1, t1 in insns[j:j + 16]:
            op = t1.split(None, 1)[0]
            if t1.startswith("s_or_b64 exec, exec, " + saved):
                # A write above the restore is harmless when it only re-establishes, for the lanes of the `then` block, a register those lanes clobbered INSIDE the
                # block (the tail of a save / restore or a rematerialised constant: the lanes that skipped still hold the value).  It is the defect when the block
                # never touches the register: then the instruction is a definition every rejoining lane needs, and the lanes that skipped do not get it.
                clobbered = set()
                for a3, t3 in insns[i + 1:j]:
                    if writes_vgpr(t3):
                        clobbered |= dest_vgprs(t3)
                for a2, t2 in pending:
                    found.append((label, a2, t2, a1, t1, "restores a register the block clobbered" if dest_vgprs(t2) <= clobbered else "DEFINES A REGISTER THE BLOCK NEVER WRITES"))
                break
            if op.startswith(STOP) or EXEC_RESTORE.match(t1) or re.match(r"^s_\w+ " + re.escape(saved) + r"\b", t1):
                break                                 # (another region starts, or the saved mask is redefined: not the simple shape)
            if writes_vgpr(t1):
                pending.append((a1, t1))
    # The same hazard at an `else`: the block that a skipped `then` lands in begins with the s_or_saveexec_b64 / s_andn2_saveexec_b64 that flips EXEC to the other lanes — a vector
    # register defined above it is defined for the wrong half of the wave.  (No build of rounds 4 / 5 has one; the shape is checked because it is the same mistake.)
    targets = set()
    for addr, text in insns:
        if text.startswith(("s_cbranch_execz", "s_cbranch_execnz")):
            off = int(text.split()[-1])
            if off < 32768:
                targets.add(addr + 4 + 4 * off)
    for label in sorted(targets):
        j = pos.get(label)
        if j is None:
            continue
        pending = []
        for a1, t1 in insns[j:j + 12]:
            op = t1.split(None, 1)[0]
            if re.match(r"^s_(or|andn2|xor)_saveexec_b64 ", t1):
                if not t1.rstrip().endswith(", -1"):         # (s_or_saveexec_b64 s[a:b], -1 opens a whole-wave section — the save / restore of an SGPR-spill register — not an else)
                    found += [(label, a2, t2, a1, t1, "DEFINES a register above the EXEC flip of an else") for a2, t2 in pending]
                break
            if op.startswith(STOP) or EXEC_RESTORE.match(t1):
                break
            if re.match(r"^(v_mov_b32_e32 v\d+, v\d+|v_mov_b64_e32 v\[|v_pk_mov_b32 |scratch_load_)", t1):     # (the allocator's products: copies and reloads — ordinary arithmetic is scheduled there legitimately)
                pending.append((a1, t1))
    return found


def scan(stream):
    name, insns, total_fn, total_insn, found = None, [], 0, 0, []
    for line in stream:
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", line)
        if m:
            if name and insns:
                found += [(name,) + f for f in check_function(name, insns)]
                total_fn += 1; total_insn += len(insns)
            name, insns = m.group(2), []
            continue
        if name is None or not line.startswith("\t") or "//" not in line:
            continue
        text, cm = line.split("//", 1)
        m = re.match(r"\s*([0-9A-Fa-f]+):", cm)
        if m:
            insns.append((int(m.group(1), 16), text.strip()))
    if name and insns:
        found += [(name,) + f for f in check_function(name, insns)]
        total_fn += 1; total_insn += len(insns)
    return total_fn, total_insn, found


def main():
    paths = sys.argv[1:] or device_objects()
    bad = 0
    for p in paths:
        d = disassembly_of(p)
        if d is None:
            print("%-40s no gfx950 code" % os.path.basename(p))
            continue
        nf, ni, found = scan(d)
        defects = [f for f in found if f[-1].startswith("DEFINES")]
        print("%-40s %4d functions, %8d instructions: %d vector-register writes above an EXEC restore, %d of them definitions the skipped lanes miss" % (os.path.basename(p), nf, ni, len(found), len(defects)))
        for name, label, a, t, ra, rt, verdict in found:
            print("    %s\n        block at %x: %x: %s   ...   %x: %s     [%s]" % (name[:110], label, a, t, ra, rt, verdict))
        bad += len(defects)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
