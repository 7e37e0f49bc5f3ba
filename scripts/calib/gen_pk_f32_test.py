#!/usr/bin/env python3
"""Generates pk_f32_modifiers.hip: every op_sel / op_sel_hi / neg_lo / neg_hi combination of v_pk_add_f32 and v_pk_mul_f32 (VGPR x VGPR and
SGPR x VGPR sources) as inline asm, checked on the device against the scalar expression the ISA defines.  Round 5: the wrong films of the big kernels
have one component of one float3 off, and the compiler's SLP vectoriser packs float3 arithmetic into these instructions with exactly such modifiers —
is an encoding it emits executed differently by gfx950?   usage: gen_pk_f32_test.py > _build/pk_f32_modifiers.hip"""
import itertools
print('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstring>\n#include <cmath>')
print('struct Case { int op, kind, os0, os1, oh0, oh1, nl0, nl1, nh0, nh1; };')
cases = []
body = []
k = 0
for op, opn in ((0, "v_pk_add_f32"), (1, "v_pk_mul_f32")):
    for kind in (0, 1):
        for os0, os1, oh0, oh1, nl0, nl1, nh0, nh1 in itertools.product((0, 1), repeat=8):
            mods = "op_sel:[%d,%d] op_sel_hi:[%d,%d] neg_lo:[%d,%d] neg_hi:[%d,%d]" % (os0, os1, oh0, oh1, nl0, nl1, nh0, nh1)
            if kind == 0:
                body.append('  { v2 r; asm volatile("%s %%0, %%1, %%2 %s" : "=v"(r) : "v"(a), "v"(b)); out[2 * %d] = r.x; out[2 * %d + 1] = r.y; }' % (opn, mods, k, k))
            else:
                body.append('  { v2 r; asm volatile("%s %%0, %%1, %%2 %s" : "=v"(r) : "s"(as), "v"(b)); out[2 * %d] = r.x; out[2 * %d + 1] = r.y; }' % (opn, mods, k, k))
            cases.append((op, kind, os0, os1, oh0, oh1, nl0, nl1, nh0, nh1))
            k += 1
print('typedef float v2 __attribute__((ext_vector_type(2)));')
print('__global__ void pk_test(const float *in, float *out) {')
print('  v2 a; a.x = in[4 * threadIdx.x]; a.y = in[4 * threadIdx.x + 1]; v2 b; b.x = in[4 * threadIdx.x + 2]; b.y = in[4 * threadIdx.x + 3];')
print('  v2 as; as.x = __builtin_amdgcn_readfirstlane(__float_as_int(in[0])) * 0 + in[0]; as.y = in[1];   /* uniform: lane 0\'s a */')
print('  as.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(in[0]))); as.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(in[1])));')
print('  out += (size_t)threadIdx.x * %d;' % (2 * k))
print("\n".join(body))
print('}')
print('static const Case cases[%d] = {%s};' % (k, ",".join("{%d,%d,%d,%d,%d,%d,%d,%d,%d,%d}" % c for c in cases)))
print('''int main() {
  const int N = %d, T = 64;
  float hin[4 * T]; for (int i = 0; i < 4 * T; ++i) hin[i] = 0.37f + 1.618f * (float)((i * 7919) %% 101) - 40.f;
  float *din, *dout; hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(float) * 2 * N * T);
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(pk_test, dim3(1), dim3(T), 0, 0, din, dout);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\\n"); return 2; }
  float *h = new float[2 * N * T]; hipMemcpy(h, dout, sizeof(float) * 2 * N * T, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < T; ++t) for (int c = 0; c < N; ++c) {
    const Case &k = cases[c];
    const float a[2] = {k.kind ? hin[0] : hin[4 * t], k.kind ? hin[1] : hin[4 * t + 1]}, b[2] = {hin[4 * t + 2], hin[4 * t + 3]};
    const float x0 = (k.nl0 ? -1.f : 1.f) * a[k.os0], x1 = (k.nl1 ? -1.f : 1.f) * b[k.os1], y0 = (k.nh0 ? -1.f : 1.f) * a[k.oh0], y1 = (k.nh1 ? -1.f : 1.f) * b[k.oh1];
    const float elo = k.op ? x0 * x1 : x0 + x1, ehi = k.op ? y0 * y1 : y0 + y1;
    const float glo = h[(size_t)t * 2 * N + 2 * c], ghi = h[(size_t)t * 2 * N + 2 * c + 1];
    if (memcmp(&elo, &glo, 4) || memcmp(&ehi, &ghi, 4)) {
      if (bad < 40) printf("MISMATCH lane %%d %%s %%s op_sel:[%%d,%%d] op_sel_hi:[%%d,%%d] neg_lo:[%%d,%%d] neg_hi:[%%d,%%d]: got (%%g, %%g) expected (%%g, %%g)\\n", t, k.op ? "v_pk_mul_f32" : "v_pk_add_f32", k.kind ? "sgpr,vgpr" : "vgpr,vgpr",
                           k.os0, k.os1, k.oh0, k.oh1, k.nl0, k.nl1, k.nh0, k.nh1, glo, ghi, elo, ehi);
      ++bad;
    }
  }
  printf("pk_f32 modifiers: %%d of %%d results differ from the ISA's definition\\n", bad, N * T);
  return bad ? 1 : 0;
}''' % k)
