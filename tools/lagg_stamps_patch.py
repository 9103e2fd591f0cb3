# Temporary instrumentation of csrc/lagg.hip: s_memtime stamps of one workgroup at its phase boundaries (the unpatched file is kept in
# /tmp/lagg_uninstr.hip: copy it back and rebuild afterwards).  python tools/lagg_stamps_patch.py && rebuild && python tools/lagg_stamps.py 256 132 16
import re,sys
p='/root/repo/eagcn_amd/csrc/lagg.hip'
s=open(p).read()
open('/tmp/lagg_uninstr.hip','w').write(s)
def rep(old,new):
    global s
    assert old in s, old[:70]
    s=s.replace(old,new,1)
rep("namespace eagcn {\n\nconstexpr int LG_CW = 32;","namespace eagcn {\n\n__device__ unsigned long long g_lagg_stamp[2][16];\n#define STAMP(i) do { if (blockIdx.x == 3 && blockIdx.y == 2 && threadIdx.x == 0) g_lagg_stamp[TRANS ? 1 : 0][i] = __builtin_readcyclecounter(); } while (0)\n\nconstexpr int LG_CW = 32;")
rep("    const eagcn_batch& bt = a.bt;\n    const int tid = threadIdx.x","    STAMP(0);\n    const eagcn_batch& bt = a.bt;\n    const int tid = threadIdx.x")
rep("        const int m0 = b0.x, R0 = b0.z, rows = min(b0.w, LAGG_RB), E0 = b1.x, ne = b1.y;","        const int m0 = b0.x, R0 = b0.z, rows = min(b0.w, LAGG_RB), E0 = b1.x, ne = b1.y;\n        if (rows > 0) STAMP(1);")
rep("        __syncthreads();                                              // B1:","        STAMP(2);\n        __syncthreads();                                              // B1:")
rep("        __syncthreads();                                              // B2:","        STAMP(4);\n        __syncthreads();                                              // B2:")
rep("        if constexpr (TRANS) __syncthreads();                         // B2b:","        STAMP(6);\n        if constexpr (TRANS) __syncthreads();                         // B2b:")
rep("        __syncthreads();                                              // B3: records and S_b / G_b are complete\n","        STAMP(7);\n        __syncthreads();                                              // B3: records and S_b / G_b are complete\n        STAMP(8);\n")
rep("        double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4]","        STAMP(11);\n        double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4]")
rep("        if (any_slow) {","        STAMP(12);\n        if (any_slow) {")
rep("    }                                                                 // (blocks)","        STAMP(9);\n    }                                                                 // (blocks)")
rep("        if (tid == 0 && h_s[256] != 0.0) atomicAdd(&out[256], h_s[256]);\n    }\n}","        if (tid == 0 && h_s[256] != 0.0) atomicAdd(&out[256], h_s[256]);\n    }\n    STAMP(10);\n}")
s=s.rstrip('\n')+'\nextern "C" int eagcn_debug_lagg_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eagcn::g_lagg_stamp), sizeof(unsigned long long) * 32); }\n'
open(p,'w').write(s)
