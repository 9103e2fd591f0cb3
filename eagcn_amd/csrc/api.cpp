// Library-level pieces of the C ABI: version, error string.
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/eagcn_hip.h"

namespace eagcn {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- event pool for fork/join between streams ------------------------------------------------------
hipEvent_t pool_event() {
    static std::vector<hipEvent_t> pool;
    static size_t next = 0;
    constexpr size_t kPool = 256;                 // far more than the forks of one step
    if (pool.empty()) {
        pool.resize(kPool, nullptr);
        for (auto& e : pool)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; }
    }
    hipEvent_t e = pool[next % kPool];
    ++next;
    return e;
}

// ---- per-kernel-class timing ---------------------------------------------------------------------
enum { PROF_NTAGS = 9 };
static const char* kTagNames[PROF_NTAGS] = {"index", "pack", "gemm", "agg", "bn", "edge", "readout", "head", "gemm_pair"};
struct ProfRec { hipEvent_t a, b; };
struct ProfState {
    bool on = false;
    std::vector<ProfRec> pool[PROF_NTAGS];
    size_t used[PROF_NTAGS] = {0};
    double work[PROF_NTAGS] = {0};
};
static ProfState g_prof;
bool prof_on() { return g_prof.on; }
void prof_begin(int tag, hipStream_t s, double work) {
    ProfState& p = g_prof;
    if (p.used[tag] == p.pool[tag].size()) {
        ProfRec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        p.pool[tag].push_back(r);
    }
    p.work[tag] += work;
    (void)hipEventRecord(p.pool[tag][p.used[tag]].a, s);
}
void prof_end(int tag, hipStream_t s) {
    ProfState& p = g_prof;
    if (p.used[tag] >= p.pool[tag].size()) return;
    (void)hipEventRecord(p.pool[tag][p.used[tag]].b, s);
    p.used[tag]++;
}
}  // namespace eagcn

extern "C" void eagcn_prof_enable(int on) { eagcn::g_prof.on = on != 0; }
extern "C" void eagcn_prof_reset(void) {
    for (int t = 0; t < eagcn::PROF_NTAGS; ++t) { eagcn::g_prof.used[t] = 0; eagcn::g_prof.work[t] = 0.0; }
}
extern "C" int eagcn_prof_ntags(void) { return eagcn::PROF_NTAGS; }
extern "C" const char* eagcn_prof_tag_name(int tag) {
    return (tag >= 0 && tag < eagcn::PROF_NTAGS) ? eagcn::kTagNames[tag] : "";
}
/* blocks until the recorded events have completed */
extern "C" int eagcn_prof_read(int tag, double* total_ms, double* work, int64_t* launches) {
    if (tag < 0 || tag >= eagcn::PROF_NTAGS) return -1;
    eagcn::ProfState& p = eagcn::g_prof;
    double ms = 0.0;
    for (size_t i = 0; i < p.used[tag]; ++i) {
        float e = 0.f;
        if (hipEventSynchronize(p.pool[tag][i].b) != hipSuccess) return -2;
        if (hipEventElapsedTime(&e, p.pool[tag][i].a, p.pool[tag][i].b) != hipSuccess) return -2;
        ms += e;
    }
    if (total_ms) *total_ms = ms;
    if (work) *work = p.work[tag];
    if (launches) *launches = (int64_t)p.used[tag];
    return 0;
}

// bumped whenever a struct of include/eagcn_hip.h changes its layout or an entry point its signature (round 4: plane fields of
// eagcn_layer_bufs, eagcn_model.fwd_signal, row capacities of the plane-GEMM entry points); eagcn_amd/_lib.py refuses another version
extern "C" int eagcn_abi_version(void) { return 7; }
/* sizeof() of the ABI structs, so a binding can verify its own struct layout (which: 0 batch,
 * 1 layout, 2 layer_params, 3 layer_bufs, 4 layer_grads) */
extern "C" size_t eagcn_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(eagcn_batch);
        case 1: return sizeof(eagcn_layout);
        case 2: return sizeof(eagcn_layer_params);
        case 3: return sizeof(eagcn_layer_bufs);
        case 4: return sizeof(eagcn_layer_grads);
        case 5: return sizeof(eagcn_head_params);
        case 6: return sizeof(eagcn_head_grads);
        case 7: return sizeof(eagcn_model);
        case 8: return sizeof(eagcn_gat_params);
        case 9: return sizeof(eagcn_pool_att);
        case 10: return sizeof(eagcn_step_loss);
        default: return 0;
    }
}
extern "C" const char* eagcn_last_error(void) { return eagcn::g_err; }
