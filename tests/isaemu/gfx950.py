"""tests/isaemu/gfx950.py — TEST-ONLY: an interpreter for the subset of the gfx950 (CDNA4, wave64) ISA the path kernels are compiled to.

It executes the BINARY: the disassembly (llvm-objdump -d) of a kernel out of the code object the product ships in libhpt.so, one wavefront at a time, 64 lanes as numpy
vectors under the EXEC mask; global memory is this process's memory (the kernel-argument block and everything it points to are built by tests/wavemu's driver:
the same DScene / RenderParams the GPU launch gets), LDS an array, scratch an array per lane.  tests/hostemu and tests/wavemu execute the SOURCE; the failures of
rounds 4 / 5 are in what the compiler made of it.  Floating point: IEEE binary32 / binary64 of numpy; v_rcp / v_rsq / v_sqrt / v_exp / v_log are correctly rounded
here (the hardware's are within 1 ulp), v_fma_f32 is a double-precision multiply-add rounded to single: the films agree with the oracle to rounding, not to the bit.
Not a model of timing, caches, or anything but architectural state."""
import bisect
import ctypes
import re
import struct
import subprocess

import numpy as np

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
U32, I32, F32, U64, I64, F64 = np.uint32, np.int32, np.float32, np.uint64, np.int64, np.float64
PRIVATE_HI, SHARED_HI = 0xFFFE0000, 0xFFFD0000        # apertures of flat addresses (src_private_base / src_shared_base)
M32 = 0xFFFFFFFF


class EmuError(RuntimeError):
    pass


def f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def d2u(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


# ---- operands -------------------------------------------------------------------------------------------------------------------
SPECIAL = {"vcc": ("s", 106, 2), "vcc_lo": ("s", 106, 1), "vcc_hi": ("s", 107, 1), "m0": ("s", 124, 1), "exec": ("exec", 0, 2), "exec_lo": ("exec", 0, 1), "exec_hi": ("exec", 1, 1),
           "scc": ("scc", 0, 1), "off": ("off", 0, 0), "null": ("null", 0, 0), "flat_scratch": ("s", 102, 2), "flat_scratch_lo": ("s", 102, 1), "flat_scratch_hi": ("s", 103, 1),
           "xnack_mask": ("s", 104, 2), "src_private_base": ("lit64", PRIVATE_HI << 32, 2), "src_shared_base": ("lit64", SHARED_HI << 32, 2),
           "src_vccz": ("vccz", 0, 1), "src_execz": ("execz", 0, 1), "src_scc": ("scc", 0, 1)}


class Op:
    __slots__ = ("kind", "idx", "n", "neg", "abs", "sext", "fval", "ival", "is_float")

    def __init__(self, kind, idx=0, n=1):
        self.kind, self.idx, self.n, self.neg, self.abs, self.sext, self.fval, self.ival, self.is_float = kind, idx, n, False, False, False, None, None, False

    def __repr__(self):
        return "%s%s%d:%d" % ("-" if self.neg else "", self.kind, self.idx, self.n)


def parse_operand(t):
    t = t.strip()
    neg = ab = sext = False
    if t.startswith("-") and not re.match(r"^-[0-9.]", t):
        neg, t = True, t[1:]
    if t.startswith("|") and t.endswith("|"):
        ab, t = True, t[1:-1]
    m = re.match(r"^(neg|abs|sext)\((.*)\)$", t)
    while m:
        if m.group(1) == "neg":
            neg = True
        elif m.group(1) == "abs":
            ab = True
        else:
            sext = True
        t = m.group(2)
        if t.startswith("|") and t.endswith("|"):
            ab, t = True, t[1:-1]
        m = re.match(r"^(neg|abs|sext)\((.*)\)$", t)
    m = re.match(r"^gpr_idx\((.*)\)$", t)
    if m:                                          # s_set_gpr_idx_on: which operands of the following VALU instructions are indexed
        o = Op("gpridx")
        o.fval = set(x.strip() for x in m.group(1).split(",") if x.strip())
        return o
    m = re.match(r"^([vsa])(\d+)$", t)
    if m:
        o = Op(m.group(1), int(m.group(2)), 1)
    else:
        m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", t)
        if m:
            o = Op(m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1)
        elif t in SPECIAL:
            k, i, n = SPECIAL[t]
            o = Op(k, i, n)
            if k == "lit64":
                o.kind, o.ival = "lit", i
        elif re.match(r"^-?(0x[0-9a-fA-F]+|\d+)$", t):
            o = Op("lit")
            o.ival = int(t, 0) & 0xFFFFFFFFFFFFFFFF if int(t, 0) >= 0 else int(t, 0)
        elif re.match(r"^-?\d+\.\d*(e[-+]?\d+)?$", t) or t in ("0.15915494", "0.15915494309189532"):
            o = Op("lit")
            o.fval, o.is_float = float(t), True
        else:
            raise EmuError("operand? %r" % t)
    o.neg, o.abs, o.sext = neg, ab, sext
    return o


MOD_RE = re.compile(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi):\[([^\]]*)\]|\s+(offset|offset0|offset1|bitop3|dst_sel|dst_unused|src0_sel|src1_sel|mul|div|format):(\S+)|\s+(glc|slc|sc0|sc1|nt|clamp|gds|lds|nv)\b"
                    r"|\s+(vmcnt|lgkmcnt|expcnt)\(\d+\)")


class Insn:
    __slots__ = ("addr", "size", "text", "op", "ops", "mods", "fn", "target")


def parse_insn(addr, size, text):
    ins = Insn()
    ins.addr, ins.size, ins.text, ins.mods, ins.fn, ins.target = addr, size, text, {}, None, None
    parts = text.split(None, 1)
    ins.op = parts[0]
    rest = " " + parts[1] if len(parts) > 1 else ""
    if ins.op in ("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_endpgm", "s_barrier", "s_waitcnt_depctr", "s_sethalt", "s_trap", "s_setreg_imm32_b32", "s_setreg_b32", "s_icache_inv", "s_dcache_wb",
                  "buffer_wbl2", "buffer_inv", "s_ttracedata", "s_code_end", "s_waitcnt_vscnt", "s_set_gpr_idx_off"):
        ins.ops = []
        return ins

    def grab(m):
        if m.group(1):
            ins.mods[m.group(1)] = [int(x) for x in m.group(2).split(",")]
        elif m.group(3):
            v = m.group(4)
            ins.mods[m.group(3)] = int(v, 0) if re.match(r"^-?(0x[0-9a-fA-F]+|\d+)$", v) else v
        elif m.group(5):
            ins.mods[m.group(5)] = True
        return ""
    rest = MOD_RE.sub(grab, rest)
    rest = rest.strip()
    ins.ops = [parse_operand(t) for t in rest.split(",")] if rest else []
    return ins


def disassemble(co_path, symbols):
    """-> list of Insn of the given functions, {addr: index}, {symbol: addr}"""
    return parse_listing(subprocess.run([OBJDUMP, "-d", "--disassemble-symbols=" + ",".join(symbols), co_path], check=True, capture_output=True, text=True).stdout)


def parse_listing(out):
    """the same from a listing (llvm-objdump -d: `<addr> <symbol>:` headers, `\ttext // ADDR: WORDS` lines)"""
    insns, index, starts = [], {}, {}
    cur = None
    for line in out.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", line)
        if m:
            cur = m.group(2)
            starts[cur] = int(m.group(1), 16)
            continue
        if cur is None or "//" not in line or not line.startswith("\t"):
            continue
        text, cm = line.split("//", 1)
        m = re.match(r"\s*([0-9A-Fa-f]+):((?:\s+[0-9A-Fa-f]{8})+)", cm)
        if not m:
            continue
        addr = int(m.group(1), 16)
        size = 4 * len(m.group(2).split())
        ins = parse_insn(addr, size, text.strip())
        index[addr] = len(insns)
        insns.append(ins)
    return insns, index, starts


def kernel_descriptor(co_path, kernel):
    """the fields of the 64-byte kernel descriptor the wave's initial state depends on"""
    syms = subprocess.run([READELF, "-sW", co_path], check=True, capture_output=True, text=True).stdout
    kd_addr = None
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[-1] == kernel + ".kd":
            kd_addr = int(f[1], 16)
    if kd_addr is None:
        raise EmuError("no descriptor for " + kernel)
    secs = subprocess.run([READELF, "-SW", co_path], check=True, capture_output=True, text=True).stdout
    m = re.search(r"\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
    ro_addr, ro_off = int(m.group(1), 16), int(m.group(2), 16)
    with open(co_path, "rb") as f:
        f.seek(ro_off + kd_addr - ro_addr)
        kd = f.read(64)
    group, private, kernarg = struct.unpack_from("<III", kd, 0)
    rsrc3, rsrc1, rsrc2 = struct.unpack_from("<III", kd, 44)
    props, = struct.unpack_from("<H", kd, 56)
    return {"group": group, "private": private, "kernarg": kernarg, "rsrc1": rsrc1, "rsrc2": rsrc2, "rsrc3": rsrc3, "props": props,
            "user_sgprs": (rsrc2 >> 1) & 31, "wg_id_x": (rsrc2 >> 7) & 1, "wg_id_y": (rsrc2 >> 8) & 1, "wg_id_z": (rsrc2 >> 9) & 1, "wg_info": (rsrc2 >> 10) & 1,
            "scratch": rsrc2 & 1, "kernarg_ptr": (props >> 3) & 1, "dispatch_ptr": (props >> 1) & 1, "queue_ptr": (props >> 2) & 1, "dispatch_id": (props >> 4) & 1,
            "private_buf": props & 1, "flat_scratch_init": (props >> 5) & 1, "private_size": (props >> 6) & 1}


# ---- memory ---------------------------------------------------------------------------------------------------------------------
class HostMemory:
    """global memory = this process's address space, guarded by /proc/self/maps so that a wild address is an error of the emulated kernel, not a crash of the test"""

    def __init__(self):
        self.refresh()

    def refresh(self):
        self.lo, self.hi = [], []
        for line in open("/proc/self/maps"):
            f = line.split()
            if len(f) >= 2 and f[1][0] == "r":
                a, b = f[0].split("-")
                self.lo.append(int(a, 16)); self.hi.append(int(b, 16))

    def check(self, addr, n):
        i = bisect.bisect_right(self.lo, addr) - 1
        if i < 0 or addr + n > self.hi[i]:
            # (adjacent mappings)
            j = i
            end = addr
            while j >= 0 and j < len(self.lo) and self.lo[j] <= end < self.hi[j]:
                end = self.hi[j]
                if end >= addr + n:
                    return
                j += 1
            self.refresh()
            i = bisect.bisect_right(self.lo, addr) - 1
            if i < 0 or addr + n > self.hi[i]:
                raise EmuError("memory access fault: %d bytes at 0x%x" % (n, addr))

    def load(self, addr, ndw):
        self.check(addr, 4 * ndw)
        return (ctypes.c_uint32 * ndw).from_address(addr)

    def store(self, addr, vals):
        self.check(addr, 4 * len(vals))
        a = (ctypes.c_uint32 * len(vals)).from_address(addr)
        for i, v in enumerate(vals):
            a[i] = int(v)


# ---- the wavefront ----------------------------------------------------------------------------------------------------------------
LANES = np.arange(64, dtype=np.uint64)
LANE_BITS = (np.uint64(1) << LANES)
SIGN32, SIGN64 = U32(0x80000000), U64(0x8000000000000000)


def mask_to_bools(m):
    return (np.uint64(m) & LANE_BITS) != 0


def bools_to_mask(b):
    return int(np.bitwise_or.reduce(np.where(b, LANE_BITS, np.uint64(0))))


def bcast(x):
    return x if isinstance(x, np.ndarray) and x.shape == (64,) else np.full(64, x)


def _fma32(a, b, c):
    return (a.astype(F64) * b.astype(F64) + c.astype(F64)).astype(F32)


def _cvt_sat(x, lo, hi, dt):
    x = np.where(np.isnan(x), 0.0, x)
    return np.clip(np.trunc(x), lo, hi).astype(dt)


def _fclass(x, mask):
    b = x.view(U32)
    e, m, s = (b >> U32(23)) & U32(0xff), b & U32(0x7fffff), (b >> U32(31)) != 0
    nan, inf, zero, den = (e == 255) & (m != 0), (e == 255) & (m == 0), (e == 0) & (m == 0), (e == 0) & (m != 0)
    norm = ~(nan | inf | zero | den)
    snan, qnan = nan & ((m & U32(0x400000)) == 0), nan & ((m & U32(0x400000)) != 0)
    cls = [snan, qnan, inf & s, norm & s, den & s, zero & s, zero & ~s, den & ~s, norm & ~s, inf & ~s]
    r = np.zeros(64, dtype=bool)
    for i, c in enumerate(cls):
        r |= c & (((mask >> U32(i)) & U32(1)) != 0)
    return r


def _min_f(a, b):
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.minimum(a, b)))


def _max_f(a, b):
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.maximum(a, b)))


CMP = {"f": lambda a, b: np.zeros(64, bool), "lt": np.less, "eq": np.equal, "le": np.less_equal, "gt": np.greater, "lg": lambda a, b: (a < b) | (a > b), "ge": np.greater_equal,
       "o": lambda a, b: ~(np.isnan(a) | np.isnan(b)), "u": lambda a, b: np.isnan(a) | np.isnan(b), "nge": lambda a, b: ~(a >= b), "nlg": lambda a, b: ~((a < b) | (a > b)),
       "ngt": lambda a, b: ~(a > b), "nle": lambda a, b: ~(a <= b), "neq": lambda a, b: ~(a == b), "nlt": lambda a, b: ~(a < b), "tru": lambda a, b: np.ones(64, bool),
       "ne": np.not_equal, "t": lambda a, b: np.ones(64, bool)}
SDWA_SEL = {"BYTE_0": (0, 0xff), "BYTE_1": (8, 0xff), "BYTE_2": (16, 0xff), "BYTE_3": (24, 0xff), "WORD_0": (0, 0xffff), "WORD_1": (16, 0xffff), "DWORD": (0, 0xffffffff)}


class Wave:
    def __init__(self, prog, mem, lds, kd, kernarg_addr, wg_id, wave_in_group, trace=None):
        self.insns, self.index = prog
        self.mem, self.lds = mem, lds
        self.V = np.zeros((512, 64), dtype=U32)
        self.S = [0] * 128
        self.exec = (1 << 64) - 1
        self.em = np.ones(64, dtype=bool)
        self.scc = 0
        self.scratch = np.zeros((64, 16384), dtype=U32)      # 64 KB of private memory per lane
        self.pc = 0
        self.done = False
        self.count = 0
        self.trace = trace
        self.gpr_idx_on, self.gpr_idx, self.gpr_idx_mode = False, 0, set()     # s_set_gpr_idx_on: relative VGPR indexing of the VALU instructions that follow
        self.pre = None          # (debugging: called with (wave, instruction) BEFORE the instruction executes)
        # initial state (AMDHSA, gfx9 family with architected flat scratch): user SGPRs in descriptor order, then the workgroup ids; v0 = packed work-item id
        i = 0
        if kd["private_buf"]:
            i += 4
        if kd["dispatch_ptr"]:
            i += 2
        if kd["queue_ptr"]:
            i += 2
        if kd["kernarg_ptr"]:
            self.S[i], self.S[i + 1] = kernarg_addr & M32, kernarg_addr >> 32
            i += 2
        if kd["dispatch_id"]:
            i += 2
        if kd["flat_scratch_init"]:
            i += 2
        if kd["private_size"]:
            i += 1
        if i != kd["user_sgprs"]:
            raise EmuError("user SGPRs: counted %d, descriptor says %d" % (i, kd["user_sgprs"]))
        if kd["wg_id_x"]:
            self.S[i] = wg_id
            i += 1
        if kd["wg_id_y"]:
            i += 1
        if kd["wg_id_z"]:
            i += 1
        self.V[0] = (np.arange(64, dtype=U32) + U32(64 * wave_in_group))

    # -- operand access --
    def set_exec(self, m):
        self.exec = m & ((1 << 64) - 1)
        self.em = mask_to_bools(self.exec)

    def s32(self, o):
        k = o.kind
        if k == "s":
            return self.S[o.idx]
        if k == "lit":
            return (f2u(o.fval) if o.is_float else o.ival) & M32
        if k == "exec":
            return (self.exec >> (32 * o.idx)) & M32
        if k == "scc":
            return self.scc
        if k == "vccz":
            return 1 if (self.S[106] | self.S[107]) == 0 else 0
        if k == "execz":
            return 1 if self.exec == 0 else 0
        if k == "null":
            return 0
        raise EmuError("scalar read of %r" % o)

    def s64(self, o):
        k = o.kind
        if k == "s":
            return self.S[o.idx] | (self.S[o.idx + 1] << 32) if o.n >= 2 else self.S[o.idx]
        if k == "lit":
            if o.is_float:
                return d2u(o.fval)
            return o.ival & 0xFFFFFFFFFFFFFFFF
        if k == "exec":
            return self.exec
        if k == "scc":
            return self.scc
        raise EmuError("scalar 64 read of %r" % o)

    def ws32(self, o, v):
        v &= M32
        if o.kind == "s":
            self.S[o.idx] = v
        elif o.kind == "exec":
            self.set_exec((self.exec & ~(M32 << (32 * o.idx))) | (v << (32 * o.idx)))
        elif o.kind == "null":
            pass
        else:
            raise EmuError("scalar write of %r" % o)

    def ws64(self, o, v):
        v &= 0xFFFFFFFFFFFFFFFF
        if o.kind == "s":
            self.S[o.idx] = v & M32
            if o.n >= 2:
                self.S[o.idx + 1] = v >> 32
        elif o.kind == "exec":
            self.set_exec(v)
        elif o.kind == "null":
            pass
        else:
            raise EmuError("scalar 64 write of %r" % o)

    def r32(self, o):
        """a 32-bit source of a vector instruction as uint32 (array or scalar), sign modifiers applied to the bits"""
        if o.kind == "v":
            x = self.V[o.idx]
        else:
            x = U32(self.s32(o))
        if o.abs:
            x = x & U32(0x7fffffff)
        if o.neg:
            x = x ^ SIGN32
        return x

    def rf(self, o):
        x = self.r32(o)
        return x.view(F32) if isinstance(x, np.ndarray) else np.full(64, x, dtype=U32).view(F32)

    def r64(self, o, fp=False):
        if o.kind == "v":
            x = self.V[o.idx].astype(U64) | (self.V[o.idx + 1].astype(U64) << U64(32))
        elif o.kind == "lit":
            if o.is_float:
                x = U64(d2u(o.fval))
            elif fp and not (-16 <= (o.ival if o.ival < (1 << 63) else o.ival - (1 << 64)) <= 64):
                x = U64((o.ival & M32) << 32)
            else:
                x = U64(o.ival & 0xFFFFFFFFFFFFFFFF)
        else:
            x = U64(self.s64(o))
        if o.abs:
            x = x & U64(0x7fffffffffffffff)
        if o.neg:
            x = x ^ SIGN64
        return x

    def rd(self, o):
        return bcast(self.r64(o, True)).view(F64)

    def w32(self, o, val):
        if o.kind != "v":
            raise EmuError("vector write to %r" % o)
        v = val.view(U32) if isinstance(val, np.ndarray) and val.dtype != U32 and val.dtype.itemsize == 4 else val
        if isinstance(v, np.ndarray) and v.dtype != U32:
            v = v.astype(U32)
        np.copyto(self.V[o.idx], v, where=self.em, casting="unsafe")

    def w64(self, o, val):
        v = bcast(val)
        v = v.view(U64) if v.dtype != U64 else v
        np.copyto(self.V[o.idx], (v & U64(M32)).astype(U32), where=self.em)
        np.copyto(self.V[o.idx + 1], (v >> U64(32)).astype(U32), where=self.em)

    def wmask(self, o, bools):
        """a lane mask result (compares, carries): bits of inactive lanes are 0"""
        m = bools_to_mask(bools & self.em)
        self.ws64(o, m)

    def rmask(self, o):
        return mask_to_bools(self.s64(o))

    # -- run --
    def run(self, max_insns=None):
        insns, n = self.insns, 0
        with np.errstate(all="ignore"):
            while not self.done:
                ins = insns[self.pc]
                self.pc += 1
                if ins.fn is None:
                    ins.fn = bind(ins)
                if self.pre is not None:
                    self.pre(self, ins)
                try:
                    if self.gpr_idx_on and ins.op.startswith("v_"):
                        self.exec_indexed(ins)
                    else:
                        ins.fn(self, ins)
                except EmuError as e:
                    raise EmuError("%s\n  at 0x%x: %s" % (e, ins.addr, ins.text)) from None
                except Exception as e:
                    raise EmuError("%s: %s\n  at 0x%x: %s" % (type(e).__name__, e, ins.addr, ins.text)) from None
                n += 1
                if self.trace is not None:
                    self.trace(self, ins)
                if max_insns is not None and n >= max_insns:
                    break
        self.count += n
        return n

    def exec_indexed(self, ins):
        """a VALU instruction under s_set_gpr_idx_on: the compiler only ever brackets v_mov_b32 (dynamic indexing of an array kept in registers)"""
        if ins.op != "v_mov_b32_e32" or ins.ops[0].kind != "v":
            raise EmuError("VGPR index mode around something else than v_mov_b32")
        d = ins.ops[0].idx + (self.gpr_idx if "DST" in self.gpr_idx_mode else 0)
        o = ins.ops[1]
        if o.kind == "v":
            src = self.V[o.idx + (self.gpr_idx if "SRC0" in self.gpr_idx_mode else 0)]
        else:
            src = U32(self.s32(o))
        if not 0 <= d < 512:
            raise EmuError("indexed VGPR write outside the register file")
        np.copyto(self.V[d], src, where=self.em)

    def jump(self, addr):
        i = self.index.get(addr)
        if i is None:
            raise EmuError("jump to 0x%x: not in the disassembled functions" % addr)
        self.pc = i


# ---- instruction semantics --------------------------------------------------------------------------------------------------------
def _nop(w, i):
    pass


def _endpgm(w, i):
    w.done = True


def _branch_target(ins):
    off = ins.ops[0].ival
    if off >= 32768:
        off -= 65536
    return ins.addr + 4 + 4 * off


def _mk_branch(cond):
    def f(w, i):
        if i.target is None:
            i.target = w.index[_branch_target(i)]
        if cond(w):
            w.pc = i.target
    return f


BRANCH = {"s_branch": lambda w: True, "s_cbranch_scc0": lambda w: w.scc == 0, "s_cbranch_scc1": lambda w: w.scc == 1, "s_cbranch_vccz": lambda w: (w.S[106] | w.S[107]) == 0,
          "s_cbranch_vccnz": lambda w: (w.S[106] | w.S[107]) != 0, "s_cbranch_execz": lambda w: w.exec == 0, "s_cbranch_execnz": lambda w: w.exec != 0}


def _sx32(v):
    return v - (1 << 32) if v & 0x80000000 else v


def _sx64(v):
    return v - (1 << 64) if v & (1 << 63) else v


def _brev32(v):
    return int("{:032b}".format(v)[::-1], 2)


def _ffbh(v, bits):      # position of the first 1 from the MSB, -1 if none
    return -1 if v == 0 else bits - v.bit_length()


def _ff1(v):
    return -1 if v == 0 else (v & -v).bit_length() - 1


M64 = 0xFFFFFFFFFFFFFFFF
# scalar ALU: name -> (width of sources, f(w, a, b) -> (result, scc or None))
SOP2 = {
    "s_add_u32": (32, lambda w, a, b: ((a + b) & M32, 1 if a + b > M32 else 0)), "s_sub_u32": (32, lambda w, a, b: ((a - b) & M32, 1 if b > a else 0)),
    "s_add_i32": (32, lambda w, a, b: ((a + b) & M32, 1 if not (-(1 << 31) <= _sx32(a) + _sx32(b) < (1 << 31)) else 0)),
    "s_sub_i32": (32, lambda w, a, b: ((a - b) & M32, 1 if not (-(1 << 31) <= _sx32(a) - _sx32(b) < (1 << 31)) else 0)),
    "s_addc_u32": (32, lambda w, a, b: ((a + b + w.scc) & M32, 1 if a + b + w.scc > M32 else 0)), "s_subb_u32": (32, lambda w, a, b: ((a - b - w.scc) & M32, 1 if b + w.scc > a else 0)),
    "s_min_i32": (32, lambda w, a, b: (a if _sx32(a) < _sx32(b) else b, 1 if _sx32(a) < _sx32(b) else 0)), "s_min_u32": (32, lambda w, a, b: (min(a, b), 1 if a < b else 0)),
    "s_max_i32": (32, lambda w, a, b: (a if _sx32(a) > _sx32(b) else b, 1 if _sx32(a) > _sx32(b) else 0)), "s_max_u32": (32, lambda w, a, b: (max(a, b), 1 if a > b else 0)),
    "s_cselect_b32": (32, lambda w, a, b: (a if w.scc else b, None)), "s_cselect_b64": (64, lambda w, a, b: (a if w.scc else b, None)),
    "s_and_b32": (32, lambda w, a, b: (a & b, int((a & b) != 0))), "s_and_b64": (64, lambda w, a, b: (a & b, int((a & b) != 0))),
    "s_or_b32": (32, lambda w, a, b: (a | b, int((a | b) != 0))), "s_or_b64": (64, lambda w, a, b: (a | b, int((a | b) != 0))),
    "s_xor_b32": (32, lambda w, a, b: (a ^ b, int((a ^ b) != 0))), "s_xor_b64": (64, lambda w, a, b: (a ^ b, int((a ^ b) != 0))),
    "s_andn2_b32": (32, lambda w, a, b: (a & ~b & M32, int((a & ~b & M32) != 0))), "s_andn2_b64": (64, lambda w, a, b: (a & ~b & M64, int((a & ~b & M64) != 0))),
    "s_orn2_b32": (32, lambda w, a, b: ((a | ~b) & M32, int(((a | ~b) & M32) != 0))), "s_orn2_b64": (64, lambda w, a, b: ((a | ~b) & M64, int(((a | ~b) & M64) != 0))),
    "s_nand_b32": (32, lambda w, a, b: (~(a & b) & M32, int((~(a & b) & M32) != 0))), "s_nand_b64": (64, lambda w, a, b: (~(a & b) & M64, int((~(a & b) & M64) != 0))),
    "s_nor_b32": (32, lambda w, a, b: (~(a | b) & M32, int((~(a | b) & M32) != 0))), "s_nor_b64": (64, lambda w, a, b: (~(a | b) & M64, int((~(a | b) & M64) != 0))),
    "s_xnor_b32": (32, lambda w, a, b: (~(a ^ b) & M32, int((~(a ^ b) & M32) != 0))), "s_xnor_b64": (64, lambda w, a, b: (~(a ^ b) & M64, int((~(a ^ b) & M64) != 0))),
    "s_lshl_b32": (32, lambda w, a, b: ((a << (b & 31)) & M32, int(((a << (b & 31)) & M32) != 0))), "s_lshr_b32": (32, lambda w, a, b: (a >> (b & 31), int((a >> (b & 31)) != 0))),
    "s_ashr_i32": (32, lambda w, a, b: ((_sx32(a) >> (b & 31)) & M32, int(((_sx32(a) >> (b & 31)) & M32) != 0))),
    "s_lshl_b64": (6432, lambda w, a, b: ((a << (b & 63)) & M64, int(((a << (b & 63)) & M64) != 0))), "s_lshr_b64": (6432, lambda w, a, b: (a >> (b & 63), int((a >> (b & 63)) != 0))),
    "s_ashr_i64": (6432, lambda w, a, b: ((_sx64(a) >> (b & 63)) & M64, int(((_sx64(a) >> (b & 63)) & M64) != 0))),
    "s_mul_i32": (32, lambda w, a, b: ((a * b) & M32, None)), "s_mul_hi_u32": (32, lambda w, a, b: ((a * b) >> 32, None)), "s_mul_hi_i32": (32, lambda w, a, b: ((_sx32(a) * _sx32(b)) >> 32 & M32, None)),
    "s_bfe_u32": (32, lambda w, a, b: ((a >> (b & 31)) & ((1 << ((b >> 16) & 0x7f)) - 1), int(((a >> (b & 31)) & ((1 << ((b >> 16) & 0x7f)) - 1)) != 0))),
    "s_lshl1_add_u32": (32, lambda w, a, b: (((a << 1) + b) & M32, 1 if (a << 1) + b > M32 else 0)), "s_lshl2_add_u32": (32, lambda w, a, b: (((a << 2) + b) & M32, 1 if (a << 2) + b > M32 else 0)),
    "s_lshl3_add_u32": (32, lambda w, a, b: (((a << 3) + b) & M32, 1 if (a << 3) + b > M32 else 0)), "s_lshl4_add_u32": (32, lambda w, a, b: (((a << 4) + b) & M32, 1 if (a << 4) + b > M32 else 0)),
    "s_pack_ll_b32_b16": (32, lambda w, a, b: ((a & 0xffff) | ((b & 0xffff) << 16), None)),
}
SOP1 = {
    "s_mov_b32": (32, lambda w, a: (a, None)), "s_mov_b64": (64, lambda w, a: (a, None)), "s_cmov_b32": (32, None), "s_cmov_b64": (64, None),
    "s_not_b32": (32, lambda w, a: (~a & M32, int((~a & M32) != 0))), "s_not_b64": (64, lambda w, a: (~a & M64, int((~a & M64) != 0))),
    "s_brev_b32": (32, lambda w, a: (_brev32(a), None)), "s_bcnt1_i32_b32": (32, lambda w, a: (bin(a).count("1"), int(a != 0))), "s_bcnt1_i32_b64": (6400, lambda w, a: (bin(a).count("1"), int(a != 0))),
    "s_bcnt0_i32_b64": (6400, lambda w, a: (64 - bin(a).count("1"), int(a != M64))),
    "s_ff1_i32_b32": (32, lambda w, a: (_ff1(a) & M32, None)), "s_ff1_i32_b64": (6400, lambda w, a: (_ff1(a) & M32, None)),
    "s_flbit_i32": (32, lambda w, a: (_ffbh((a ^ M32) if a & 0x80000000 else a, 32) & M32, None)),      # first bit that differs from the sign bit, from the MSB (-1: none)
    "s_flbit_i32_b32": (32, lambda w, a: (_ffbh(a, 32) & M32, None)), "s_flbit_i32_b64": (6400, lambda w, a: (_ffbh(a, 64) & M32, None)),
    "s_sext_i32_i8": (32, lambda w, a: (((a & 0xff) ^ 0x80) - 0x80 & M32, None)), "s_sext_i32_i16": (32, lambda w, a: (((a & 0xffff) ^ 0x8000) - 0x8000 & M32, None)),
    "s_abs_i32": (32, lambda w, a: (abs(_sx32(a)) & M32, int(a != 0))),
    "s_bitset1_b32": (None, None), "s_bitset0_b32": (None, None),
}
SAVEEXEC = {"s_and_saveexec_b64": lambda e, s: s & e, "s_or_saveexec_b64": lambda e, s: s | e, "s_xor_saveexec_b64": lambda e, s: s ^ e, "s_andn2_saveexec_b64": lambda e, s: s & ~e & M64,
            "s_orn2_saveexec_b64": lambda e, s: (s | ~e) & M64, "s_nand_saveexec_b64": lambda e, s: ~(s & e) & M64, "s_nor_saveexec_b64": lambda e, s: ~(s | e) & M64,
            "s_xnor_saveexec_b64": lambda e, s: ~(s ^ e) & M64, "s_andn1_saveexec_b64": lambda e, s: ~s & e & M64, "s_orn1_saveexec_b64": lambda e, s: (~s | e) & M64}
SCMP = {"eq": lambda a, b: a == b, "lg": lambda a, b: a != b, "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b, "lt": lambda a, b: a < b, "le": lambda a, b: a <= b}


def bind_scalar(ins):
    op, ops = ins.op, ins.ops
    if op in SOP2:
        width, f = SOP2[op]
        rd_a = (lambda w, o: w.s64(o)) if width in (64, 6432) else (lambda w, o: w.s32(o))
        rd_b = (lambda w, o: w.s64(o)) if width == 64 else (lambda w, o: w.s32(o))
        wr = (lambda w, o, v: w.ws64(o, v)) if width in (64, 6432) else (lambda w, o, v: w.ws32(o, v))

        def f2(w, i):
            r, scc = f(w, rd_a(w, ops[1]), rd_b(w, ops[2]))
            wr(w, ops[0], r)
            if scc is not None:
                w.scc = scc
        return f2
    if op in ("s_cmov_b32", "s_cmov_b64"):
        def fc(w, i):
            if w.scc:
                (w.ws64 if op.endswith("64") else w.ws32)(ops[0], (w.s64 if op.endswith("64") else w.s32)(ops[1]))
        return fc
    if op in ("s_bitset1_b32", "s_bitset0_b32"):
        def fb(w, i):
            d, b = w.s32(ops[0]), 1 << (w.s32(ops[1]) & 31)
            w.ws32(ops[0], d | b if op == "s_bitset1_b32" else d & ~b)
        return fb
    if op in SOP1:
        width, f = SOP1[op]
        rd = (lambda w, o: w.s64(o)) if width in (64, 6400) else (lambda w, o: w.s32(o))
        wr = (lambda w, o, v: w.ws64(o, v)) if width == 64 else (lambda w, o, v: w.ws32(o, v))

        def f1(w, i):
            r, scc = f(w, rd(w, ops[1]))
            wr(w, ops[0], r)
            if scc is not None:
                w.scc = scc
        return f1
    if op in SAVEEXEC:
        f = SAVEEXEC[op]

        def fs(w, i):
            e = w.exec
            src = w.s64(ops[1])
            w.ws64(ops[0], e)
            w.set_exec(f(e, src))
            w.scc = int(w.exec != 0)
        return fs
    if op == "s_movk_i32":
        return lambda w, i: w.ws32(ops[0], ((ops[1].ival & 0xffff) ^ 0x8000) - 0x8000)
    if op == "s_addk_i32":
        def fak(w, i):
            a, b = _sx32(w.s32(ops[0])), ((ops[1].ival & 0xffff) ^ 0x8000) - 0x8000
            w.ws32(ops[0], a + b)
            w.scc = 0 if -(1 << 31) <= a + b < (1 << 31) else 1
        return fak
    if op == "s_mulk_i32":
        return lambda w, i: w.ws32(ops[0], _sx32(w.s32(ops[0])) * (((ops[1].ival & 0xffff) ^ 0x8000) - 0x8000))
    m = re.match(r"^s_cmp(k?)_(eq|lg|gt|ge|lt|le)_([iu])(32|64)$", op)
    if m:
        cmp, signed, wide, k = SCMP[m.group(2)], m.group(3) == "i", m.group(4) == "64", m.group(1) == "k"

        def fcmp(w, i):
            a = w.s64(ops[0]) if wide else w.s32(ops[0])
            if k:
                b = ops[1].ival & 0xffff
                b = ((b ^ 0x8000) - 0x8000) & M32 if signed else b
            else:
                b = w.s64(ops[1]) if wide else w.s32(ops[1])
            if signed:
                a, b = (_sx64(a), _sx64(b)) if wide else (_sx32(a), _sx32(b))
            w.scc = int(cmp(a, b))
        return fcmp
    if op in ("s_bitcmp0_b32", "s_bitcmp1_b32"):
        want = 1 if op == "s_bitcmp1_b32" else 0
        return lambda w, i: setattr(w, "scc", int(((w.s32(ops[0]) >> (w.s32(ops[1]) & 31)) & 1) == want))
    if op == "s_set_gpr_idx_on":
        def fon(w, i):
            w.gpr_idx_on, w.gpr_idx, w.gpr_idx_mode = True, w.s32(ops[0]) & 0xff, ops[1].fval
        return fon
    if op == "s_set_gpr_idx_off":
        return lambda w, i: setattr(w, "gpr_idx_on", False)
    if op == "s_set_gpr_idx_idx":
        return lambda w, i: setattr(w, "gpr_idx", w.s32(ops[0]) & 0xff)
    if op == "s_getpc_b64":
        return lambda w, i: w.ws64(ops[0], i.addr + 4)
    if op == "s_setpc_b64":
        return lambda w, i: w.jump(w.s64(ops[0]))
    if op == "s_swappc_b64":
        def fsw(w, i):
            t = w.s64(ops[1])
            w.ws64(ops[0], i.addr + 4)
            w.jump(t)
        return fsw
    m = re.match(r"^s_load_dword(x(\d+))?$", op)
    if m:
        n = int(m.group(2) or 1)

        def fl(w, i):
            base = w.s64(ops[1])
            off = ops[2].ival if ops[2].kind == "lit" else w.s32(ops[2])
            off += i.mods.get("offset", 0)
            a = w.mem.load((base + off) & ~3, n)
            for j in range(n):
                w.S[ops[0].idx + j] = a[j]
        return fl
    if op in ("s_store_dword", "s_atomic_add"):
        raise EmuError("scalar stores are not used by the path kernels")
    return None


# ---- vector ALU ---------------------------------------------------------------------------------------------------------------------
def _arr(x):
    return x if isinstance(x, np.ndarray) else np.full(64, x, dtype=U32)


def _f(x):
    return _arr(x).view(F32)


def _i(x):
    return _arr(x).view(I32)


def _u64(x):
    return _arr(x).astype(U64)


def _frexp_mant(x):
    m, e = np.frexp(x)
    return np.where(np.isfinite(x), m, x).astype(F32)


def _frexp_exp(x):
    m, e = np.frexp(x)
    return np.where(np.isfinite(x) & (x != 0), e, 0).astype(I32)


def _rcp(x):
    return (F32(1.0) / x).astype(F32)


def _sqrt(x):
    return np.sqrt(x).astype(F32)


def _ffbh_u32(a):
    a = _arr(a)
    r = np.full(64, -1, dtype=I32)
    nz = a != 0
    lg = np.floor(np.log2(a[nz].astype(F64))).astype(I32)
    # (float64 represents every uint32 exactly: floor(log2) is the index of the top bit)
    r[nz] = 31 - lg
    return r


def _ffbl_b32(a):
    a = _arr(a)
    low = a & (~a + U32(1))
    r = np.full(64, -1, dtype=I32)
    nz = a != 0
    r[nz] = np.log2(low[nz].astype(F64)).astype(I32)
    return r


def _bfrev(a):
    a = _arr(a).copy()
    a = ((a >> U32(1)) & U32(0x55555555)) | ((a & U32(0x55555555)) << U32(1))
    a = ((a >> U32(2)) & U32(0x33333333)) | ((a & U32(0x33333333)) << U32(2))
    a = ((a >> U32(4)) & U32(0x0f0f0f0f)) | ((a & U32(0x0f0f0f0f)) << U32(4))
    a = ((a >> U32(8)) & U32(0x00ff00ff)) | ((a & U32(0x00ff00ff)) << U32(8))
    return (a >> U32(16)) | (a << U32(16))


def _popc(a):
    a = _arr(a).copy()
    a = a - ((a >> U32(1)) & U32(0x55555555))
    a = (a & U32(0x33333333)) + ((a >> U32(2)) & U32(0x33333333))
    a = (a + (a >> U32(4))) & U32(0x0f0f0f0f)
    return (a * U32(0x01010101)) >> U32(24)


def _bitop3(a, b, c, tbl):
    a, b, c = _arr(a), _arr(b), _arr(c)
    r = np.zeros(64, dtype=U32)
    for idx in range(8):
        if (tbl >> idx) & 1:
            t = (a if idx & 4 else ~a) & (b if idx & 2 else ~b) & (c if idx & 1 else ~c)
            r |= t
    return r


def _ldexp(x, e):
    return np.ldexp(x.astype(F64), np.clip(_i(e), -300, 300)).astype(F32)


def _med3(a, b, c):
    return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))


# D = f(S0, S1[, S2]) on 32-bit lanes; sources arrive as uint32 (array or numpy scalar)
VOP = {
    "v_mov_b32": lambda a: a, "v_not_b32": lambda a: ~_arr(a),
    "v_add_f32": lambda a, b: _f(a) + _f(b), "v_sub_f32": lambda a, b: _f(a) - _f(b), "v_subrev_f32": lambda a, b: _f(b) - _f(a), "v_mul_f32": lambda a, b: _f(a) * _f(b),
    "v_min_f32": lambda a, b: _min_f(_f(a), _f(b)), "v_max_f32": lambda a, b: _max_f(_f(a), _f(b)),
    "v_fma_f32": lambda a, b, c: _fma32(_f(a), _f(b), _f(c)), "v_mad_f32": lambda a, b, c: _f(a) * _f(b) + _f(c),
    "v_med3_f32": lambda a, b, c: _med3(_f(a), _f(b), _f(c)), "v_min3_f32": lambda a, b, c: _min_f(_min_f(_f(a), _f(b)), _f(c)), "v_max3_f32": lambda a, b, c: _max_f(_max_f(_f(a), _f(b)), _f(c)),
    "v_rcp_f32": lambda a: _rcp(_f(a)), "v_rcp_iflag_f32": lambda a: _rcp(_f(a)), "v_rsq_f32": lambda a: (F32(1.0) / np.sqrt(_f(a).astype(F64))).astype(F32), "v_sqrt_f32": lambda a: _sqrt(_f(a)),
    "v_exp_f32": lambda a: np.exp2(_f(a).astype(F64)).astype(F32), "v_log_f32": lambda a: np.log2(_f(a).astype(F64)).astype(F32),
    "v_sin_f32": lambda a: np.sin(_f(a).astype(F64) * (2 * np.pi)).astype(F32), "v_cos_f32": lambda a: np.cos(_f(a).astype(F64) * (2 * np.pi)).astype(F32),
    "v_rndne_f32": lambda a: np.rint(_f(a)), "v_trunc_f32": lambda a: np.trunc(_f(a)), "v_floor_f32": lambda a: np.floor(_f(a)), "v_ceil_f32": lambda a: np.ceil(_f(a)),
    "v_fract_f32": lambda a: _f(a) - np.floor(_f(a)),
    "v_frexp_mant_f32": lambda a: _frexp_mant(_f(a)), "v_frexp_exp_i32_f32": lambda a: _frexp_exp(_f(a)), "v_ldexp_f32": lambda a, b: _ldexp(_f(a), b),
    "v_cvt_f32_i32": lambda a: _i(a).astype(F32), "v_cvt_f32_u32": lambda a: _arr(a).astype(F32),
    "v_cvt_i32_f32": lambda a: _cvt_sat(_f(a).astype(F64), -2147483648.0, 2147483647.0, I64).astype(I32), "v_cvt_u32_f32": lambda a: _cvt_sat(_f(a).astype(F64), 0.0, 4294967295.0, U64).astype(U32),
    "v_cvt_rpi_i32_f32": lambda a: _cvt_sat(np.floor(_f(a).astype(F64) + 0.5), -2147483648.0, 2147483647.0, I64).astype(I32),
    "v_cvt_flr_i32_f32": lambda a: _cvt_sat(np.floor(_f(a).astype(F64)), -2147483648.0, 2147483647.0, I64).astype(I32),
    "v_cvt_f32_ubyte0": lambda a: (_arr(a) & U32(0xff)).astype(F32), "v_cvt_f32_ubyte1": lambda a: ((_arr(a) >> U32(8)) & U32(0xff)).astype(F32),
    "v_cvt_f32_ubyte2": lambda a: ((_arr(a) >> U32(16)) & U32(0xff)).astype(F32), "v_cvt_f32_ubyte3": lambda a: (_arr(a) >> U32(24)).astype(F32),
    "v_add_u32": lambda a, b: _arr(a) + _arr(b), "v_sub_u32": lambda a, b: _arr(a) - _arr(b), "v_subrev_u32": lambda a, b: _arr(b) - _arr(a),
    "v_add_i32": lambda a, b: _arr(a) + _arr(b), "v_sub_i32": lambda a, b: _arr(a) - _arr(b),
    "v_mul_lo_u32": lambda a, b: _arr(a) * _arr(b), "v_mul_hi_u32": lambda a, b: ((_u64(a) * _u64(b)) >> U64(32)).astype(U32),
    "v_mul_hi_i32": lambda a, b: ((_i(a).astype(I64) * _i(b).astype(I64)) >> I64(32)).astype(I32),
    "v_mul_u32_u24": lambda a, b: (_arr(a) & U32(0xffffff)) * (_arr(b) & U32(0xffffff)), "v_mul_i32_i24": lambda a, b: (((_i(a) << I32(8)) >> I32(8)) * ((_i(b) << I32(8)) >> I32(8))),
    "v_mad_u32_u24": lambda a, b, c: (_arr(a) & U32(0xffffff)) * (_arr(b) & U32(0xffffff)) + _arr(c),
    "v_mad_i32_i24": lambda a, b, c: (((_i(a) << I32(8)) >> I32(8)) * ((_i(b) << I32(8)) >> I32(8))) + _i(c),
    "v_and_b32": lambda a, b: _arr(a) & _arr(b), "v_or_b32": lambda a, b: _arr(a) | _arr(b), "v_xor_b32": lambda a, b: _arr(a) ^ _arr(b), "v_xnor_b32": lambda a, b: ~(_arr(a) ^ _arr(b)),
    "v_bfm_b32": lambda a, b: (((U32(1) << (_arr(a) & U32(31))) - U32(1)) << (_arr(b) & U32(31))).astype(U32),      # bitfield mask: ((1 << S0[4:0]) - 1) << S1[4:0]
    "v_lshlrev_b32": lambda a, b: _arr(b) << (_arr(a) & U32(31)), "v_lshrrev_b32": lambda a, b: _arr(b) >> (_arr(a) & U32(31)), "v_ashrrev_i32": lambda a, b: _i(b) >> (_arr(a) & U32(31)).view(I32),
    "v_min_u32": lambda a, b: np.minimum(_arr(a), _arr(b)), "v_max_u32": lambda a, b: np.maximum(_arr(a), _arr(b)), "v_min_i32": lambda a, b: np.minimum(_i(a), _i(b)), "v_max_i32": lambda a, b: np.maximum(_i(a), _i(b)),
    "v_min3_u32": lambda a, b, c: np.minimum(np.minimum(_arr(a), _arr(b)), _arr(c)), "v_max3_u32": lambda a, b, c: np.maximum(np.maximum(_arr(a), _arr(b)), _arr(c)),
    "v_min3_i32": lambda a, b, c: np.minimum(np.minimum(_i(a), _i(b)), _i(c)), "v_max3_i32": lambda a, b, c: np.maximum(np.maximum(_i(a), _i(b)), _i(c)),
    "v_med3_i32": lambda a, b, c: _med3(_i(a), _i(b), _i(c)), "v_med3_u32": lambda a, b, c: _med3(_arr(a), _arr(b), _arr(c)),
    "v_add3_u32": lambda a, b, c: _arr(a) + _arr(b) + _arr(c), "v_lshl_add_u32": lambda a, b, c: (_arr(a) << (_arr(b) & U32(31))) + _arr(c), "v_add_lshl_u32": lambda a, b, c: (_arr(a) + _arr(b)) << (_arr(c) & U32(31)),
    "v_lshl_or_b32": lambda a, b, c: (_arr(a) << (_arr(b) & U32(31))) | _arr(c), "v_and_or_b32": lambda a, b, c: (_arr(a) & _arr(b)) | _arr(c), "v_or3_b32": lambda a, b, c: _arr(a) | _arr(b) | _arr(c),
    "v_xad_u32": lambda a, b, c: (_arr(a) ^ _arr(b)) + _arr(c),
    "v_bfe_u32": lambda a, b, c: (_arr(a) >> (_arr(b) & U32(31))) & ((U64(1) << (_u64(c) & U64(31))) - U64(1)).astype(U32),
    "v_bfe_i32": lambda a, b, c: np.where((_arr(c) & U32(31)) == 0, I32(0), (_i(a) << ((U32(32) - (_arr(c) & U32(31)) - (_arr(b) & U32(31))) & U32(31)).view(I32)) >> ((U32(32) - (_arr(c) & U32(31))) & U32(31)).view(I32)),
    "v_bfi_b32": lambda a, b, c: (_arr(a) & _arr(b)) | (~_arr(a) & _arr(c)),
    "v_alignbit_b32": lambda a, b, c: (((_u64(a) << U64(32)) | _u64(b)) >> (_u64(c) & U64(31))).astype(U32),
    "v_alignbyte_b32": lambda a, b, c: (((_u64(a) << U64(32)) | _u64(b)) >> (U64(8) * (_u64(c) & U64(3)))).astype(U32),
    "v_perm_b32": None,
    "v_ffbh_u32": lambda a: _ffbh_u32(a), "v_ffbh_i32": lambda a: _ffbh_u32(np.where((_arr(a) & U32(0x80000000)) != 0, ~_arr(a), _arr(a))), "v_ffbl_b32": lambda a: _ffbl_b32(a), "v_bfrev_b32": lambda a: _bfrev(a), "v_bcnt_u32_b32": lambda a, b: _popc(a) + _arr(b),
    "v_mbcnt_lo_u32_b32": lambda a, b: _popc(_arr(a) & ((U64(1) << np.minimum(LANES, U64(32))) - U64(1)).astype(U32)) + _arr(b),
    "v_mbcnt_hi_u32_b32": lambda a, b: _popc(_arr(a) & ((U64(1) << (np.maximum(LANES, U64(32)) - U64(32))) - U64(1)).astype(U32)) + _arr(b),
    "v_sad_u32": lambda a, b, c: np.where(_arr(a) > _arr(b), _arr(a) - _arr(b), _arr(b) - _arr(a)) + _arr(c),
    "v_lshrrev_b16": lambda a, b: ((_arr(b) & U32(0xffff)) >> (_arr(a) & U32(15))), "v_lshlrev_b16": lambda a, b: ((_arr(b) << (_arr(a) & U32(15))) & U32(0xffff)),
    "v_add_u16": lambda a, b: (_arr(a) + _arr(b)) & U32(0xffff), "v_and_b16": lambda a, b: _arr(a) & _arr(b) & U32(0xffff),
}
FLOAT_RESULT = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_min_f32", "v_max_f32", "v_fma_f32", "v_mad_f32", "v_med3_f32", "v_min3_f32", "v_max3_f32", "v_rcp_f32", "v_rcp_iflag_f32",
                "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rndne_f32", "v_trunc_f32", "v_floor_f32", "v_ceil_f32", "v_fract_f32", "v_ldexp_f32", "v_frexp_mant_f32", "v_cvt_f32_i32",
                "v_cvt_f32_u32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_div_fixup_f32", "v_div_fmas_f32", "v_cvt_f32_f64", "v_sin_f32", "v_cos_f32"}
# 64-bit: (kinds of the sources: 'd' double, 'q' uint64, 'w' 32-bit), result kind, f
VOP64 = {
    "v_add_f64": ("dd", "d", lambda a, b: a + b), "v_mul_f64": ("dd", "d", lambda a, b: a * b), "v_fma_f64": ("ddd", "d", lambda a, b, c: a * b + c),
    "v_min_f64": ("dd", "d", _min_f), "v_max_f64": ("dd", "d", _max_f),
    "v_rcp_f64": ("d", "d", lambda a: 1.0 / a), "v_sqrt_f64": ("d", "d", np.sqrt), "v_rsq_f64": ("d", "d", lambda a: 1.0 / np.sqrt(a)),
    "v_floor_f64": ("d", "d", np.floor), "v_ceil_f64": ("d", "d", np.ceil), "v_trunc_f64": ("d", "d", np.trunc), "v_rndne_f64": ("d", "d", np.rint), "v_fract_f64": ("d", "d", lambda a: a - np.floor(a)),
    "v_ldexp_f64": ("dw", "d", lambda a, b: np.ldexp(a, np.clip(_i(b), -3000, 3000))),
    "v_frexp_mant_f64": ("d", "d", lambda a: np.where(np.isfinite(a), np.frexp(a)[0], a)), "v_frexp_exp_i32_f64": ("d", "w", lambda a: np.where(np.isfinite(a) & (a != 0), np.frexp(a)[1], 0).astype(I32)),
    "v_cvt_f64_f32": ("w", "d", lambda a: _f(a).astype(F64)), "v_cvt_f32_f64": ("d", "w", lambda a: a.astype(F32)), "v_cvt_f64_i32": ("w", "d", lambda a: _i(a).astype(F64)), "v_cvt_f64_u32": ("w", "d", lambda a: _arr(a).astype(F64)),
    "v_cvt_i32_f64": ("d", "w", lambda a: _cvt_sat(a, -2147483648.0, 2147483647.0, I64).astype(I32)), "v_cvt_u32_f64": ("d", "w", lambda a: _cvt_sat(a, 0.0, 4294967295.0, U64).astype(U32)),
    "v_mov_b64": ("q", "q", lambda a: a), "v_lshlrev_b64": ("wq", "q", lambda a, b: b << (_u64(a) & U64(63))), "v_lshrrev_b64": ("wq", "q", lambda a, b: b >> (_u64(a) & U64(63))),
    "v_ashrrev_i64": ("wq", "q", lambda a, b: (b.view(I64) >> (_u64(a) & U64(63)).view(I64)).view(U64)),
    "v_lshl_add_u64": ("qwq", "q", lambda a, b, c: (a << (_u64(b) & U64(7))) + c),
    "v_and_b64": ("qq", "q", lambda a, b: a & b), "v_or_b64": ("qq", "q", lambda a, b: a | b), "v_xor_b64": ("qq", "q", lambda a, b: a ^ b),
}


def _read_kind(w, o, k):
    if k == "d":
        return w.rd(o)
    if k == "q":
        return bcast(w.r64(o))
    return w.r32(o)


def _omod(ins, r):
    if "mul" in ins.mods:
        r = r * F32(ins.mods["mul"])
    if "div" in ins.mods:
        r = r / F32(ins.mods["div"])
    if ins.mods.get("clamp"):
        r = np.clip(np.where(np.isnan(r), 0.0, r), 0.0, 1.0).astype(r.dtype)
    return r


def _sdwa_src(x, sel, sext):
    sh, m = SDWA_SEL[sel]
    x = (_arr(x) >> U32(sh)) & U32(m)
    if sext and m != 0xffffffff:
        bits = 8 if m == 0xff else 16
        x = ((x.view(I32) << I32(32 - bits)) >> I32(32 - bits)).view(U32)
    return x


def bind_vector(ins):
    op, ops, mods = ins.op, ins.ops, ins.mods
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    sdwa = op.endswith("_sdwa")
    # ---- compares ----
    m = re.match(r"^v_cmp(x?)_(\w+?)_(f32|f64|i32|u32|i64|u64|u16|i16)$", base)
    if m and m.group(2) in CMP:
        cmp, ty, x = CMP[m.group(2)], m.group(3), m.group(1) == "x"
        explicit = len(ops) == 3
        dst = ops[0] if explicit else SPEC_VCC
        sa, sb = (ops[1], ops[2]) if explicit else (ops[0], ops[1])

        def fcmp(w, i):
            if ty in ("f64", "i64", "u64"):
                a, b = (w.rd(sa), w.rd(sb)) if ty == "f64" else (bcast(w.r64(sa)), bcast(w.r64(sb)))
                if ty == "i64":
                    a, b = a.view(I64), b.view(I64)
            else:
                a, b = w.r32(sa), w.r32(sb)
                if sdwa:
                    a, b = _sdwa_src(a, mods.get("src0_sel", "DWORD"), sa.sext), _sdwa_src(b, mods.get("src1_sel", "DWORD"), sb.sext)
                a, b = _arr(a), _arr(b)
                if ty == "f32":
                    a, b = a.view(F32), b.view(F32)
                elif ty in ("i32", "i16"):
                    a, b = a.view(I32), b.view(I32)
            r = cmp(a, b)
            w.wmask(dst, r)
            if x:
                w.set_exec(w.s64(dst))
        return fcmp
    m = re.match(r"^v_cmp(x?)_class_(f32|f64)$", base)
    if m:
        explicit = len(ops) == 3
        dst = ops[0] if explicit else SPEC_VCC
        sa, sb = (ops[1], ops[2]) if explicit else (ops[0], ops[1])
        if m.group(2) == "f64":
            raise EmuError("v_cmp_class_f64")
        return lambda w, i: w.wmask(dst, _fclass(w.rf(sa), _arr(w.r32(sb))))
    # ---- select, carries ----
    if base == "v_cndmask_b32":
        sel = ops[3] if len(ops) == 4 else SPEC_VCC

        def fcnd(w, i):
            a, b = w.r32(ops[1]), w.r32(ops[2])
            if sdwa:
                a, b = _sdwa_src(a, mods.get("src0_sel", "DWORD"), ops[1].sext), _sdwa_src(b, mods.get("src1_sel", "DWORD"), ops[2].sext)
            w.w32(ops[0], np.where(w.rmask(sel), _arr(b), _arr(a)))
        return fcnd
    m = re.match(r"^v_(add|sub|subrev)_co_u32$", base)
    if m:
        kind = m.group(1)

        def fco(w, i):
            a, b = _u64(w.r32(ops[2])), _u64(w.r32(ops[3]))
            if kind == "subrev":
                a, b = b, a
            r = a + b if kind == "add" else a - b
            carry = (r >> U64(32)) != 0
            w.w32(ops[0], (r & U64(M32)).astype(U32))
            w.wmask(ops[1], carry)
        return fco
    m = re.match(r"^v_(addc|subb|subbrev)_co_u32$", base)
    if m:
        kind = m.group(1)

        def fcc(w, i):
            a, b, c = _u64(w.r32(ops[2])), _u64(w.r32(ops[3])), w.rmask(ops[4]).astype(U64)
            if kind == "subbrev":
                a, b = b, a
            r = a + b + c if kind == "addc" else a - b - c
            w.w32(ops[0], (r & U64(M32)).astype(U32))
            w.wmask(ops[1], (r >> U64(32)) != 0)
        return fcc
    if base in ("v_mad_u64_u32", "v_mad_i64_i32"):
        signed = base == "v_mad_i64_i32"

        def fmad(w, i):
            a, b, c = _arr(w.r32(ops[2])), _arr(w.r32(ops[3])), bcast(w.r64(ops[4]))
            p = (a.view(I32).astype(I64) * b.view(I32).astype(I64)).view(U64) if signed else a.astype(U64) * b.astype(U64)
            r = p + c
            w.w64(ops[0], r)
            w.wmask(ops[1], r < c)
        return fmad
    # ---- fused multiply-add forms with an implicit or literal operand ----
    if base == "v_fmac_f32":
        return lambda w, i: w.w32(ops[0], _omod(i, _fma32(w.rf(ops[1]), w.rf(ops[2]), w.V[ops[0].idx].view(F32))))
    if base == "v_mac_f32":
        return lambda w, i: w.w32(ops[0], w.rf(ops[1]) * w.rf(ops[2]) + w.V[ops[0].idx].view(F32))
    if base == "v_fmaak_f32":
        return lambda w, i: w.w32(ops[0], _fma32(w.rf(ops[1]), w.rf(ops[2]), w.rf(ops[3])))
    if base == "v_fmamk_f32":
        return lambda w, i: w.w32(ops[0], _fma32(w.rf(ops[1]), w.rf(ops[2]), w.rf(ops[3])))      # D = S0 * K + S1: the assembler prints (S0, K, S1)
    if base == "v_fmac_f64":
        return lambda w, i: w.w64(ops[0], (w.rd(ops[1]) * w.rd(ops[2]) + bcast(w.r64(ops[0])).view(F64)).view(U64))
    if base == "v_bitop3_b32":
        return lambda w, i: w.w32(ops[0], _bitop3(w.r32(ops[1]), w.r32(ops[2]), w.r32(ops[3]), mods["bitop3"]))
    if base == "v_bitop3_b16":
        return lambda w, i: w.w32(ops[0], (_bitop3(w.r32(ops[1]), w.r32(ops[2]), w.r32(ops[3]), mods["bitop3"]) & U32(0xffff)) | (w.V[ops[0].idx] & U32(0xffff0000)))
    # ---- division support ----
    if base == "v_div_scale_f32":
        def fds(w, i):
            s0, s1, s2 = w.rf(ops[2]), w.rf(ops[3]), w.rf(ops[4])
            e1, e2 = ((s1.view(U32) >> U32(23)) & U32(0xff)).astype(I32), ((s2.view(U32) >> U32(23)) & U32(0xff)).astype(I32)
            den1 = (e1 == 0) & (s1 != 0)
            rcp_den = np.abs(F64(1.0) / s1.astype(F64)) < 1.1754943508222875e-38
            quo_den = np.abs(s2.astype(F64) / s1.astype(F64)) < 1.1754943508222875e-38
            same01, same02 = s0.view(U32) == s1.view(U32), s0.view(U32) == s2.view(U32)
            up, dn = np.ldexp(s0.astype(F64), 64).astype(F32), np.ldexp(s0.astype(F64), -64).astype(F32)
            d, vcc = s0.copy(), np.zeros(64, bool)
            done = np.zeros(64, bool)
            c = (s2 == 0) | (s1 == 0)
            d = np.where(c, F32(np.nan), d); done |= c
            c = ~done & (e2 - e1 >= 96)
            vcc |= c; d = np.where(c & same01, up, d); done |= c
            c = ~done & den1
            d = np.where(c, up, d); done |= c
            c = ~done & rcp_den & quo_den
            vcc |= c; d = np.where(c & same01, up, d); done |= c
            c = ~done & rcp_den
            d = np.where(c, dn, d); done |= c
            c = ~done & quo_den
            vcc |= c; d = np.where(c & same02, up, d); done |= c
            c = ~done & (e2 <= 23)
            d = np.where(c, up, d)
            w.w32(ops[0], d)
            w.wmask(ops[1], vcc)
        return fds
    if base == "v_div_fmas_f32":
        def fdf(w, i):
            r = _fma32(w.rf(ops[1]), w.rf(ops[2]), w.rf(ops[3]))
            w.w32(ops[0], np.where(w.rmask(SPEC_VCC), np.ldexp(r.astype(F64), 32).astype(F32), r))
        return fdf
    if base == "v_div_fixup_f32":
        def fdx(w, i):
            q, den, num = w.rf(ops[1]), w.rf(ops[2]), w.rf(ops[3])
            sign = ((den.view(U32) ^ num.view(U32)) & SIGN32) != 0
            sgn = lambda v: np.where(sign, -v, v).astype(F32)   # noqa: E731
            r = sgn(np.abs(q))
            en, ed = ((num.view(U32) >> U32(23)) & U32(0xff)).astype(I32), ((den.view(U32) >> U32(23)) & U32(0xff)).astype(I32)
            r = np.where(en - ed < -150, sgn(np.zeros(64, F32)), r)
            r = np.where(ed == 255, r, r)
            r = np.where(np.isinf(den) | (num == 0), sgn(np.zeros(64, F32)), r)
            r = np.where((den == 0) | np.isinf(num), sgn(np.full(64, np.inf, F32)), r)
            r = np.where(((den == 0) & (num == 0)) | (np.isinf(den) & np.isinf(num)), F32(np.nan), r)
            r = np.where(np.isnan(den), den, r)
            r = np.where(np.isnan(num), num, r)
            # Where v_div_scale scaled an operand (denormal operands or quotient, quotients near the top of the range) the sequence's scaling is not
            # modelled exactly (v_div_fmas' direction): there the correctly rounded quotient stands in.  Everywhere else the result is the sequence's.
            scaled = (en - ed >= 96) | (ed == 0) | (en <= 23) | (ed >= 253) | (en - ed <= -126)
            exact = (num.astype(F64) / den.astype(F64)).astype(F32)
            r = np.where(scaled & np.isfinite(num) & np.isfinite(den) & (den != 0) & (num != 0), exact, r)
            w.w32(ops[0], r.astype(F32))
        return fdx
    # ---- packed f32 / moves ----
    m = re.match(r"^v_pk_(add|mul|fma)_f32$", base)
    if m:
        kind, ns = m.group(1), 3 if m.group(1) == "fma" else 2
        osel, oselh = mods.get("op_sel", [0] * ns), mods.get("op_sel_hi", [1] * ns)
        nlo, nhi = mods.get("neg_lo", [0] * ns), mods.get("neg_hi", [0] * ns)

        def fpk(w, i):
            los, his = [], []
            for k in range(ns):
                o = ops[1 + k]
                if o.kind == "v":
                    lo, hi = w.V[o.idx], w.V[o.idx + 1]
                elif o.kind == "lit":
                    lo, hi = U32(w.s32(o)), U32(0)
                else:
                    v = w.s64(o)
                    lo, hi = U32(v & M32), U32(v >> 32)
                a, b = _f(hi if osel[k] else lo), _f(hi if oselh[k] else lo)
                los.append(-a if nlo[k] else a); his.append(-b if nhi[k] else b)
            if kind == "add":
                rl, rh = los[0] + los[1], his[0] + his[1]
            elif kind == "mul":
                rl, rh = los[0] * los[1], his[0] * his[1]
            else:
                rl, rh = _fma32(los[0], los[1], los[2]), _fma32(his[0], his[1], his[2])
            np.copyto(w.V[ops[0].idx], rl.view(U32), where=w.em)
            np.copyto(w.V[ops[0].idx + 1], rh.view(U32), where=w.em)
        return fpk
    if base == "v_pk_mov_b32":
        osel, oselh = mods.get("op_sel", [0, 0]), mods.get("op_sel_hi", [1, 1])

        def fpm(w, i):
            def half(o, hi):
                if o.kind == "v":
                    return w.V[o.idx + 1] if hi else w.V[o.idx]
                v = w.s64(o)
                return np.full(64, (v >> 32) if hi else (v & M32), dtype=U32)
            lo, hi = half(ops[1], osel[0]).copy(), half(ops[2], osel[1]).copy()      # D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]] (a 64-bit move is op_sel:[0,1])
            np.copyto(w.V[ops[0].idx], lo, where=w.em)
            np.copyto(w.V[ops[0].idx + 1], hi, where=w.em)
        return fpm
    if base == "v_swap_b32":          # D <-> S0 in the active lanes (two VGPRs)
        def fsw(w, i):
            a, b = w.V[ops[0].idx].copy(), w.V[ops[1].idx].copy()
            np.copyto(w.V[ops[0].idx], b, where=w.em)
            np.copyto(w.V[ops[1].idx], a, where=w.em)
        return fsw
    # ---- lanes ----
    if base == "v_readlane_b32":
        return lambda w, i: w.ws32(ops[0], int(w.V[ops[1].idx][w.s32(ops[2]) & 63]))
    if base == "v_readfirstlane_b32":
        def frf(w, i):
            l = _ff1(w.exec) if w.exec else 0
            w.ws32(ops[0], int(w.V[ops[1].idx][l]) if ops[1].kind == "v" else w.s32(ops[1]))
        return frf
    if base == "v_writelane_b32":
        def fwl(w, i):
            w.V[ops[0].idx][w.s32(ops[2]) & 63] = w.s32(ops[1])
        return fwl
    # ---- 64-bit ----
    if base in VOP64:
        kinds, rk, f = VOP64[base]

        def f64op(w, i):
            r = f(*[_read_kind(w, ops[1 + k], kinds[k]) for k in range(len(kinds))])
            if rk == "w":
                w.w32(ops[0], r)
            else:
                if rk == "d":
                    r = _omod(i, r) if (i.mods.get("clamp") or "mul" in i.mods) else r
                w.w64(ops[0], bcast(r).astype(F64).view(U64) if rk == "d" else r)
        return f64op
    # ---- the table ----
    f = VOP.get(base)
    if f is not None:
        ns = f.__code__.co_argcount
        is_float = base in FLOAT_RESULT
        if sdwa:
            dsel, dun = mods.get("dst_sel", "DWORD"), mods.get("dst_unused", "UNUSED_PAD")

            def fsd(w, i):
                srcs = [_sdwa_src(w.r32(ops[1 + k]), mods.get("src%d_sel" % k, "DWORD"), ops[1 + k].sext) for k in range(ns)]
                r = _arr(f(*srcs))
                r = r.view(U32) if r.dtype.itemsize == 4 else r.astype(U32)
                sh, m_ = SDWA_SEL[dsel]
                if m_ != 0xffffffff:
                    keep = w.V[ops[0].idx] & ~U32(m_ << sh) if dun == "UNUSED_PRESERVE" else U32(0)
                    r = ((r & U32(m_)) << U32(sh)) | keep
                w.w32(ops[0], r)
            return fsd
        if is_float and (mods.get("clamp") or "mul" in mods or "div" in mods):
            return lambda w, i: w.w32(ops[0], _omod(i, _arr(f(*[w.r32(ops[1 + k]) for k in range(ns)])).astype(F32)))
        if ns == 1:
            return lambda w, i: w.w32(ops[0], _arr(f(w.r32(ops[1]))))
        if ns == 2:
            return lambda w, i: w.w32(ops[0], _arr(f(w.r32(ops[1]), w.r32(ops[2]))))
        return lambda w, i: w.w32(ops[0], _arr(f(w.r32(ops[1]), w.r32(ops[2]), w.r32(ops[3]))))
    return None


SPEC_VCC = Op("s", 106, 2)


# ---- memory instructions ------------------------------------------------------------------------------------------------------------
def _simm(v, bits=13):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def bind_memory(ins):
    op, ops, mods = ins.op, ins.ops, ins.mods
    off = mods.get("offset", 0)
    m = re.match(r"^(global|flat|scratch)_(load|store)_(dword|dwordx2|dwordx3|dwordx4|ubyte|sbyte|ushort|sshort|byte|short)$", op)
    if m:
        space, kind, ty = m.groups()
        ndw = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}.get(ty, 1)
        sub = ty if ndw == 1 and ty != "dword" else None
        if kind == "load":
            dst, vaddr, saddr = ops[0], ops[1], ops[2] if len(ops) > 2 else None
        else:
            vaddr, data, saddr = ops[0], ops[1], ops[2] if len(ops) > 2 else None

        def addresses(w):
            """-> ('scratch', per-lane byte offsets) or ('global', per-lane addresses as python ints)"""
            if space == "scratch":
                base = 0 if saddr is None or saddr.kind == "off" else w.s32(saddr)
                v = 0 if vaddr.kind == "off" else w.V[vaddr.idx].astype(I64)
                return "scratch", (np.zeros(64, I64) + v + base + off)
            if saddr is not None and saddr.kind != "off":
                a = (np.zeros(64, U64) + U64(w.s64(saddr)) + w.V[vaddr.idx].astype(U64)).view(I64) + I64(off)
            else:
                a = bcast(w.r64(vaddr)).view(I64) + I64(off)
            if space == "flat":
                hi = (a.view(U64) >> U64(32)).astype(U32)
                if np.any(hi[w.em] == U32(PRIVATE_HI)):
                    if not np.all(hi[w.em] == U32(PRIVATE_HI)):
                        raise EmuError("a flat access that is private for some lanes only")
                    return "scratch", (a.view(U64) & U64(M32)).view(I64)
                if np.any(hi[w.em] == U32(SHARED_HI)):
                    raise EmuError("flat access to LDS")
            return "global", a

        def fmem(w, i):
            where, a = addresses(w)
            lanes = np.nonzero(w.em)[0]
            if where == "scratch":
                if np.any(a[lanes] & 3) and sub is None:
                    raise EmuError("unaligned scratch access")
                if np.any(a[lanes] < 0) or np.any(a[lanes] + 4 * ndw > 4 * w.scratch.shape[1]):
                    raise EmuError("scratch access outside the lane's 64 KB: offsets %s" % sorted(set(int(x) for x in a[lanes]))[:4])
                idx = a >> 2
                if sub is not None:
                    raise EmuError("sub-dword scratch access")
                for k in range(ndw):
                    if kind == "load":
                        w.V[dst.idx + k][lanes] = w.scratch[lanes, idx[lanes] + k]
                    else:
                        w.scratch[lanes, idx[lanes] + k] = w.V[data.idx + k][lanes]
                return
            mem = w.mem
            if kind == "load":
                for l in lanes:
                    ad = int(a[l])
                    if sub is None:
                        v = mem.load(ad, ndw)
                        for k in range(ndw):
                            w.V[dst.idx + k][l] = v[k]
                    else:
                        mem.check(ad, 2 if "short" in sub else 1)
                        if "short" in sub:
                            v = ctypes.c_uint16.from_address(ad).value
                            v = ((v ^ 0x8000) - 0x8000) & M32 if sub[0] == "s" else v
                        else:
                            v = ctypes.c_uint8.from_address(ad).value
                            v = ((v ^ 0x80) - 0x80) & M32 if sub[0] == "s" else v
                        w.V[dst.idx][l] = v
            else:
                for l in lanes:
                    ad = int(a[l])
                    if sub is None:
                        mem.store(ad, [w.V[data.idx + k][l] for k in range(ndw)])
                    elif sub == "byte":
                        mem.check(ad, 1); ctypes.c_uint8.from_address(ad).value = int(w.V[data.idx][l]) & 0xff
                    else:
                        mem.check(ad, 2); ctypes.c_uint16.from_address(ad).value = int(w.V[data.idx][l]) & 0xffff
        return fmem
    m = re.match(r"^(global|flat)_atomic_(add_f32|add|add_x2|umin|umax|smin|smax|and|or|xor|swap|cmpswap|inc|dec|add_f64|pk_add_f16|min_f64|max_f64)$", op)
    if m:
        kind = m.group(2)
        ret = bool(mods.get("glc") or mods.get("sc0"))
        if ret:
            dst, vaddr, data, saddr = ops[0], ops[1], ops[2], ops[3] if len(ops) > 3 else None
        else:
            dst, vaddr, data, saddr = None, ops[0], ops[1], ops[2] if len(ops) > 2 else None

        def fat(w, i):
            if saddr is not None and saddr.kind != "off":
                a = (np.zeros(64, U64) + U64(w.s64(saddr)) + w.V[vaddr.idx].astype(U64)).view(I64) + I64(off)
            else:
                a = bcast(w.r64(vaddr)).view(I64) + I64(off)
            for l in np.nonzero(w.em)[0]:
                ad = int(a[l])
                if kind == "add_f32":
                    w.mem.check(ad, 4)
                    c = ctypes.c_float.from_address(ad)
                    old = c.value
                    c.value = float(F32(old) + w.V[data.idx].view(F32)[l])
                    if ret:
                        w.V[dst.idx][l] = f2u(old)
                elif kind == "add_x2":
                    w.mem.check(ad, 8)
                    c = ctypes.c_uint64.from_address(ad)
                    old = c.value
                    c.value = (old + (int(w.V[data.idx][l]) | (int(w.V[data.idx + 1][l]) << 32))) & M64
                    if ret:
                        w.V[dst.idx][l], w.V[dst.idx + 1][l] = old & M32, old >> 32
                elif kind in ("add", "umin", "umax", "or", "and", "xor", "swap"):
                    w.mem.check(ad, 4)
                    c = ctypes.c_uint32.from_address(ad)
                    old, d = c.value, int(w.V[data.idx][l])
                    c.value = {"add": (old + d) & M32, "umin": min(old, d), "umax": max(old, d), "or": old | d, "and": old & d, "xor": old ^ d, "swap": d}[kind]
                    if ret:
                        w.V[dst.idx][l] = old
                else:
                    raise EmuError("atomic " + kind)
        return fat
    # ---- LDS ----
    if op.startswith("ds_"):
        lds = None

        def chk(w, idx, lanes):
            if np.any(idx[lanes] < 0) or np.any(idx[lanes] >= w.lds.shape[0]):
                raise EmuError("LDS access outside the workgroup's allocation: byte offsets %s" % sorted(set(int(x) * 4 for x in idx[lanes]))[-3:])
        m = re.match(r"^ds_(read|write)_b(32|64|96|128)$", op)
        if m:
            ndw, rd = int(m.group(2)) // 32, m.group(1) == "read"

            def fds(w, i):
                lanes = np.nonzero(w.em)[0]
                a = w.V[ops[1 if rd else 0].idx].astype(I64) + off
                if np.any(a[lanes] & 3):
                    raise EmuError("unaligned LDS access")
                idx = a >> 2
                chk(w, idx + ndw - 1, lanes); chk(w, idx, lanes)
                for k in range(ndw):
                    if rd:
                        w.V[ops[0].idx + k][lanes] = w.lds[idx[lanes] + k]
                    else:
                        w.lds[idx[lanes] + k] = w.V[ops[1].idx + k][lanes]
            return fds
        m = re.match(r"^ds_(read|write)2(st64)?_b(32|64)$", op)
        if m:
            rd, st, ndw = m.group(1) == "read", 64 if m.group(2) else 1, int(m.group(3)) // 32
            o0, o1 = mods.get("offset0", 0), mods.get("offset1", 0)

            def fds2(w, i):
                lanes = np.nonzero(w.em)[0]
                base = w.V[ops[1 if rd else 0].idx].astype(I64)
                for j, o in enumerate((o0, o1)):
                    idx = (base + o * st * 4 * ndw) >> 2
                    chk(w, idx, lanes); chk(w, idx + ndw - 1, lanes)
                    for k in range(ndw):
                        if rd:
                            w.V[ops[0].idx + j * ndw + k][lanes] = w.lds[idx[lanes] + k]
                        else:
                            w.lds[idx[lanes] + k] = w.V[ops[1 + j].idx + k][lanes]
            return fds2
        m = re.match(r"^ds_(min|max|add|sub|or|and|xor)(_rtn)?_(u32|i32|b32)$", op)
        if m:
            kind, rtn, ty = m.group(1), bool(m.group(2)), m.group(3)

            def fda(w, i):
                addr_op, data_op = (ops[1], ops[2]) if rtn else (ops[0], ops[1])
                for l in np.nonzero(w.em)[0]:           # (lane order: what an atomic unit would serialise)
                    ix = (int(w.V[addr_op.idx][l]) + off) >> 2
                    if not 0 <= ix < w.lds.shape[0]:
                        raise EmuError("LDS atomic outside the allocation")
                    old, d = int(w.lds[ix]), int(w.V[data_op.idx][l])
                    if ty == "i32":
                        so, sd = _sx32(old), _sx32(d)
                        new = {"min": min(so, sd), "max": max(so, sd), "add": so + sd, "sub": so - sd}[kind] & M32
                    else:
                        new = {"min": min(old, d), "max": max(old, d), "add": (old + d) & M32, "sub": (old - d) & M32, "or": old | d, "and": old & d, "xor": old ^ d}[kind]
                    w.lds[ix] = new
                    if rtn:
                        w.V[ops[0].idx][l] = old
            return fda
        if op in ("ds_bpermute_b32", "ds_permute_b32"):
            fwd = op == "ds_permute_b32"

            def fbp(w, i):
                sel = ((w.V[ops[1].idx].astype(I64) + off) >> 2) & 63
                data = w.V[ops[2].idx]
                if fwd:
                    r = w.V[ops[0].idx].copy()
                    for l in np.nonzero(w.em)[0]:
                        r[sel[l]] = data[l]
                    w.V[ops[0].idx][:] = r
                else:
                    src_active = w.em[sel]
                    w.w32(ops[0], np.where(src_active, data[sel], U32(0)))
            return fbp
        if op == "ds_swizzle_b32":
            raise EmuError("ds_swizzle")
    return None


def bind(ins):
    op = ins.op
    if op in ("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_barrier", "s_waitcnt_depctr", "s_icache_inv", "s_dcache_wb", "buffer_wbl2", "buffer_inv", "s_waitcnt_vscnt", "s_setreg_imm32_b32", "s_ttracedata"):
        return _nop
    if op == "s_endpgm":
        return _endpgm
    if op in BRANCH:
        return _mk_branch(BRANCH[op])
    f = None
    if op.startswith("s_"):
        f = bind_scalar(ins)
    elif op.startswith("v_"):
        f = bind_vector(ins)
    else:
        f = bind_memory(ins)
    if f is None:
        raise EmuError("instruction not implemented: " + ins.text)
    return f


def unimplemented(insns):
    """the instructions of a program this interpreter does not know (before anything runs)"""
    bad = {}
    for ins in insns:
        try:
            bind(ins)
        except EmuError as e:
            bad.setdefault(ins.op, str(e))
        except Exception as e:
            bad.setdefault(ins.op, "%s: %s (%s)" % (type(e).__name__, e, ins.text))
    return bad
