"""tests/isaemu/slicer.py — TEST-ONLY debugging aid: a dynamic backward slice through an interpreted wavefront.

For ONE lane it records, for every instruction that executes with that lane active (and every scalar instruction), which event last wrote each source — VGPRs, SGPRs,
the lane's scratch dwords, LDS words — and the values it read; `explain` then walks from a register or a scratch slot back through its producers.  This is how the
instruction that hands a wrong value on was found in a binary whose source is right (profiles/r05_isaemu_root_cause.md)."""
import re

import numpy as np

F32, U32 = np.float32, np.uint32


class Slicer:
    def __init__(self, lane, window=400000):
        self.lane, self.window = lane, window
        self.events = {}          # id -> (addr, text, [(name, value, producer id)], exec)
        self.last = {}            # location -> id of the event that last wrote it
        self.n = 0
        self._pending = None

    def _regs(self, o):
        if o.kind in ("v", "s"):
            return [(o.kind, o.idx + k) for k in range(o.n)]
        if o.kind == "exec":
            return [("exec", 0)]
        return []

    def _val(self, w, loc):
        k, i = loc
        if k == "v":
            return int(w.V[i][self.lane])
        if k == "s":
            return w.S[i]
        if k == "exec":
            return w.exec
        if k == "scr":
            return int(w.scratch[self.lane][i])
        if k == "lds":
            return int(w.lds[i]) if 0 <= i < w.lds.shape[0] else -1
        return 0

    def pre(self, w, ins):
        op, ops = ins.op, ins.ops
        self._pending = None
        scalar = op.startswith("s_")
        if not scalar and not ((w.exec >> self.lane) & 1) and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return
        if not ops:
            return
        srcs, dsts = [], []
        off = ins.mods.get("offset", 0)
        m = re.match(r"^scratch_(load|store)_dword(x(\d))?$", op)
        if m:
            n = int(m.group(3) or 1)
            if m.group(1) == "load":
                vaddr, saddr = ops[1], ops[2]
                base = (0 if vaddr.kind == "off" else int(w.V[vaddr.idx][self.lane])) + (0 if saddr.kind == "off" else w.S[saddr.idx]) + off
                srcs = self._regs(vaddr) + self._regs(saddr) + [("scr", (base >> 2) + k) for k in range(n)]
                dsts = [("v", ops[0].idx + k) for k in range(n)]
            else:
                vaddr, data, saddr = ops[0], ops[1], ops[2]
                base = (0 if vaddr.kind == "off" else int(w.V[vaddr.idx][self.lane])) + (0 if saddr.kind == "off" else w.S[saddr.idx]) + off
                srcs = self._regs(vaddr) + self._regs(saddr) + [("v", data.idx + k) for k in range(n)]
                dsts = [("scr", (base >> 2) + k) for k in range(n)]
        elif op in ("ds_read_b32", "ds_read_b64", "ds_read_b128"):
            n = int(op[9:]) // 32
            a = (int(w.V[ops[1].idx][self.lane]) + off) >> 2
            srcs = self._regs(ops[1]) + [("lds", a + k) for k in range(n)]
            dsts = [("v", ops[0].idx + k) for k in range(n)]
        elif op in ("ds_write_b32", "ds_write_b64", "ds_write_b128"):
            n = int(op[10:]) // 32
            a = (int(w.V[ops[0].idx][self.lane]) + off) >> 2
            srcs = self._regs(ops[0]) + [("v", ops[1].idx + k) for k in range(n)]
            dsts = [("lds", a + k) for k in range(n)]
        elif op in ("ds_write2st64_b32", "ds_read2st64_b32"):
            base = int(w.V[(ops[0] if "write" in op else ops[1]).idx][self.lane])
            locs = [("lds", (base + ins.mods.get(k, 0) * 256) >> 2) for k in ("offset0", "offset1")]
            if "write" in op:
                srcs, dsts = self._regs(ops[0]) + self._regs(ops[1]) + self._regs(ops[2]), locs
            else:
                srcs, dsts = self._regs(ops[1]) + locs, [("v", ops[0].idx), ("v", ops[0].idx + 1)]
        elif op.startswith(("global_store", "flat_store", "global_atomic")):
            for o in ops:
                srcs += self._regs(o)
        elif op.startswith("s_cbranch") or op in ("s_branch", "s_waitcnt", "s_nop"):
            return
        else:
            first_src = 1
            dsts = self._regs(ops[0])
            if re.match(r"^v_(add|sub|subrev|addc|subb|subbrev)_co_u32|^v_div_scale|^v_mad_[ui]64", op):
                dsts += self._regs(ops[1]); first_src = 2
            if op.startswith(("v_fmac", "v_mac", "v_writelane")):
                srcs += self._regs(ops[0])
            for o in ops[first_src:]:
                srcs += self._regs(o)
            if op.endswith("_e32") and op.startswith(("v_cndmask", "v_addc", "v_subb", "v_div_fmas")) or op.startswith("v_div_fmas"):
                srcs += [("s", 106), ("s", 107)]
            if "saveexec" in op:
                srcs.append(("exec", 0)); dsts.append(("exec", 0))
        self._pending = (ins, [(loc, self._val(w, loc), self.last.get(loc)) for loc in srcs], dsts)

    def post(self, w, ins):
        if self._pending is None or self._pending[0] is not ins:
            return
        _, srcs, dsts = self._pending
        self.n += 1
        self.events[self.n] = (ins.addr, ins.text, srcs, [(loc, self._val(w, loc)) for loc in dsts])
        for loc in dsts:
            self.last[loc] = self.n
        if len(self.events) > self.window:
            for k in range(self.n - len(self.events) + 1, self.n - self.window + 1):
                self.events.pop(k, None)

    @staticmethod
    def fmt(loc, v):
        k, i = loc
        name = {"v": "v%d", "s": "s%d", "scr": "scratch[%d]", "lds": "lds[%d]", "exec": "exec%d"}[k] % (i * 4 if k in ("scr", "lds") else i)
        if k == "exec":
            return "%s=%016x" % (name, v)
        f = float(np.array([v & 0xffffffff], dtype=U32).view(F32)[0])
        return "%s=%08x(%.6g)" % (name, v & 0xffffffff, f)

    def explain(self, loc, depth=4, out=None, indent=0, seen=None, follow=None):
        """prints the producers of `loc`, `depth` levels back; follow(loc, value) -> bool restricts which sources are expanded"""
        seen = seen if seen is not None else set()
        lines = out if out is not None else []
        eid = self.last.get(loc) if not isinstance(loc, int) else loc
        if eid is None or eid not in self.events:
            lines.append("  " * indent + "(no producer recorded)")
            return lines
        addr, text, srcs, dsts = self.events[eid]
        lines.append("  " * indent + "#%d %x: %s   -> %s" % (eid, addr, text, ", ".join(self.fmt(l, v) for l, v in dsts)))
        if eid in seen or depth == 0:
            return lines
        seen.add(eid)
        for l, v, p in srcs:
            if follow is not None and not follow(l, v):
                lines.append("  " * (indent + 1) + "[" + self.fmt(l, v) + "]")
                continue
            lines.append("  " * (indent + 1) + self.fmt(l, v) + (" from:" if p else " (never written: initial state)"))
            if p:
                self.explain(p, depth - 1, lines, indent + 2, seen, follow)
        return lines
