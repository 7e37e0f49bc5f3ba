#!/usr/bin/env python3
"""Which kernels did a rebuild change?  Compares, unit by unit and function by function, the gfx950 code objects of two builds (instruction text of llvm-objdump -d: the
same instructions with the same relative branch offsets are the same code wherever the function landed) and the host-side .text of the objects.  Round 5 used it to show that
a one-instantiation change to the kernel template left every other kernel of every unit exactly what the GPU had validated.

    python scripts/compare_builds.py <dir with the old hpt_kernels*.o> [<dir with the new ones> = pbrt-v2_amd/build]"""
import glob
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def code_object(obj, tmp):
    os.makedirs(tmp, exist_ok=True)
    fat, co = os.path.join(tmp, os.path.basename(obj) + ".fat"), os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    return co


def device_functions(co):
    out = subprocess.Popen([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, text=True).stdout
    res, name, h, n = {}, None, None, 0
    for l in out:
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", l)
        if m:
            if name:
                res[name] = (h.hexdigest(), n)
            name, h, n = m.group(1), hashlib.sha1(), 0
        elif name and l.startswith("\t"):
            h.update(l.split("//")[0].strip().encode())
            n += 1
    if name:
        res[name] = (h.hexdigest(), n)
    return res


def host_text(obj, tmp):
    out = os.path.join(tmp, os.path.basename(obj) + ".host")
    subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.text", obj, out])
    return hashlib.sha1(open(out, "rb").read()).hexdigest()


def main():
    old_dir = sys.argv[1]
    new_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "pbrt-v2_amd", "build")
    changed = 0
    for old in sorted(glob.glob(os.path.join(old_dir, "hpt_kernels*.o"))):
        new = os.path.join(new_dir, os.path.basename(old))
        a, b = device_functions(code_object(old, "/tmp/compare_builds/old")), device_functions(code_object(new, "/tmp/compare_builds/new"))
        diff = sorted(k for k in set(a) | set(b) if a.get(k) != b.get(k))
        host_same = host_text(old, "/tmp/compare_builds/old") == host_text(new, "/tmp/compare_builds/new")
        print("%-28s %3d device functions, %d differ; host code %s" % (os.path.basename(old), len(a), len(diff), "identical" if host_same else "DIFFERS"))
        for k in diff:
            print("    %s: %s -> %s instructions" % (k, a.get(k, ("", "-"))[1], b.get(k, ("", "-"))[1]))
        changed += len(diff)
    print("%d device functions differ in all" % changed)


if __name__ == "__main__":
    main()
