# Temporary instrumentation of csrc/head.hip: 100 MHz stamps of every wait_flag_kernel (start of the poll, end of the poll) in a ring, so that
# an UNPROFILED run shows (a) how long the step's first launch waits for its batch and (b) the period of the step graphs.
# The unpatched file is kept in /tmp/head_uninstr.hip.   python tools/flag_stamps_patch.py && rebuild && python tools/flag_stamps.py
p='/root/repo/eagcn_amd/csrc/head.hip'
s=open(p).read()
open('/tmp/head_uninstr.hip','w').write(s)
def rep(old,new):
    global s
    assert old in s, old[:70]
    s=s.replace(old,new,1)
rep("__global__ void fwd_signal_kernel(uint32_t* __restrict__ word) {\n    if (threadIdx.x == 0) ", "__device__ unsigned long long g_flag_stamp[512][4];\n__device__ unsigned g_flag_n;\n__global__ void fwd_signal_kernel(uint32_t* __restrict__ word) {\n    if (threadIdx.x == 0) g_flag_stamp[(g_flag_n - 1u) & 511u][2] = wall_clock64();\n    if (threadIdx.x == 0) ")
rep("        const unsigned long long t0 = wall_clock64();                   // 100 MHz\n", "        const unsigned long long t0 = wall_clock64();                   // 100 MHz\n        const unsigned slot = atomicAdd(&g_flag_n, 1u) & 511u;\n        g_flag_stamp[slot][0] = t0;\n")
rep("        __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n", "        __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n        g_flag_stamp[slot][1] = wall_clock64();\n")
s=s.rstrip('\n')+'\nextern "C" int eagcn_debug_flag_stamps(unsigned long long* out, unsigned* n) { if (hipMemcpyFromSymbol(n, HIP_SYMBOL(eagcn::g_flag_n), 4) != hipSuccess) return 1; return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eagcn::g_flag_stamp), sizeof(unsigned long long) * 2048); }\n'
open(p,'w').write(s)
