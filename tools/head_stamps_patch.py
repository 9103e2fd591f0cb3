# Temporary instrumentation of csrc/head2.hip: s_memtime stamps of workgroup 1 / thread 0 of head_mid_kernel at its phase boundaries
# (the unpatched file is kept in /tmp/head2_uninstr.hip: copy it back and rebuild afterwards).
#   python tools/head_stamps_patch.py && rebuild && python tools/head_stamps.py
p='/root/repo/eagcn_amd/csrc/head2.hip'
s=open(p).read()
open('/tmp/head2_uninstr.hip','w').write(s)
def rep(old,new):
    global s
    assert old in s, old[:70]
    s=s.replace(old,new,1)
rep("enum { HT_SC = 0,", "__device__ unsigned long long g_head_stamp[16];\n#define HSTAMP(i) do { if (blockIdx.x == 1 && threadIdx.x == 0) g_head_stamp[i] = __builtin_readcyclecounter(); } while (0)\n\nenum { HT_SC = 0,")
rep("    const int K = a.f3.K, N = a.f3.N, Kp = (K + 3) & ~3, Np = (N + 3) & ~3;\n    f32x4* red", "    HSTAMP(0);\n    const int K = a.f3.K, N = a.f3.N, Kp = (K + 3) & ~3, Np = (N + 3) & ~3;\n    f32x4* red")
rep("    const LossRegs R = head_loss_load(a.L, Bl, N, rb);", "    HSTAMP(1);\n    const LossRegs R = head_loss_load(a.L, Bl, N, rb);")
rep("    hf_table(a.f3, tabA, rb == 0, true);\n    hb_table(a.b3, tabB, false);\n    const float inv = 1.0f / cnt;\n    __syncthreads();", "    HSTAMP(2);\n    hf_table(a.f3, tabA, rb == 0, true);\n    hb_table(a.b3, tabB, false);\n    const float inv = 1.0f / cnt;\n    HSTAMP(3);\n    __syncthreads();\n    HSTAMP(4);")
rep("    hf_tile<VEC3, false>(f, 0, tabA, red, acc);\n    if (threadIdx.x < 64) {\n        hf_store(f, 0, acc);", "    hf_tile<VEC3, false>(f, 0, tabA, red, acc);\n    HSTAMP(5);\n    if (threadIdx.x < 64) {\n        hf_store(f, 0, acc);")
rep("    __syncthreads();\n    head_loss_block(", "    __syncthreads();\n    HSTAMP(6);\n    head_loss_block(")
rep("    __syncthreads();                                                    // d out of these rows is in LDS", "    __syncthreads();                                                    // d out of these rows is in LDS\n    HSTAMP(7);")
rep("        hb_tile<VEC3, false>(g, kb, tabB, red, tabA, Kp);\n        __syncthreads();\n    }\n}", "        hb_tile<VEC3, false>(g, kb, tabB, red, tabA, Kp);\n        __syncthreads();\n    }\n    HSTAMP(8);\n}")
s=s.rstrip('\n')+'\nextern "C" int eagcn_debug_head_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eagcn::g_head_stamp), sizeof(unsigned long long) * 16); }\n'
open(p,'w').write(s)
