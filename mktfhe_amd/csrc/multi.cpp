// Multi-device evaluator behind the C ABI (include/mktfhe.h, mkt_multi_*): ONE caller process, ONE read-only scheme, the
// batch cut into contiguous shards over several GPUs.
//
// The reference's caller is one Julia process whose threads share one read-only scheme object and allocate all scratch per
// call (README.md:38-44; src/tfhe/bootstrapping.jl:38-45); SURVEY.md 8(e) maps that onto a node as "one context and one
// stream set per device, results concatenated, keys replicated per GPU, no collective, optional one-time device-to-device
// key copy".  That is what this file is:
//   * shard i runs on HIP device devices[i]; a device may be named more than once (logical shards: forked contexts over the
//     ONE key set of that device, mkt_ctx_fork) -- which is also how the path is tested on a one-GPU box;
//   * keys are uploaded and pre-transformed ONCE, on the first device, and mkt_multi_replicate copies the resident tables to
//     every other device with hipMemcpyPeer (xGMI), then forks the logical shards;
//   * a batch call cuts [0, B) into contiguous balanced slices (the first B mod n shards hold one more), one host thread
//     per shard drives that shard's context on its own stream, and every shard writes its slice of the caller's ONE output
//     array: host memory directly, device memory directly when the array lives on the shard's device, through a
//     peer-copied staging buffer otherwise.  No data-path collective, no RCCL.
// Calls are synchronous: on return every shard's work is complete (the caller's own stream must have produced the inputs:
// mkt_multi_gate_batch synchronises the device of a device-resident input first).
#include <hip/hip_runtime.h>

#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "host_internal.h"

struct mkt_multi {
    mkt_params p;
    int arith = 0;
    std::vector<int> devices;          // per shard
    std::vector<mkt_ctx *> ctx;        // per shard; ctx[i] is a root context for the first shard of a device, a fork otherwise
    std::vector<int> root_of;          // per shard: index of the first shard on the same device
    bool stage_always = false;         // MKT_MULTI_STAGE_ALWAYS
    bool no_peer = false;              // MKT_MULTI_NO_PEER: every device-to-device copy goes through a host buffer (what happens where hipMemcpyPeer is refused)
    bool sealed = false;               // keys replicated, logical shards forked
    std::string err;
    // staging buffers of the shards whose device is not the one a device-resident argument lives on: [shard][argument slot],
    // grown on demand and kept (a hipMalloc / hipFree pair per call costs more than the peer copy of a ciphertext slice)
    struct Stage { void *p = nullptr; size_t cap = 0; };
    std::vector<std::vector<Stage>> stage;
};

namespace {
thread_local std::string g_multi_create_error;

int mfail(mkt_multi *m, int code, const std::string &msg) { if (m) m->err = msg; else g_multi_create_error = msg; return code; }

struct Slice { size_t lo, hi; };
// contiguous balanced slices of [0, B): the same rule as mktfhe_amd.distributed.shard_slices (the multi-process path)
std::vector<Slice> slices(size_t B, size_t n) {
    std::vector<Slice> out(n);
    const size_t base = B / n, rem = B % n;
    size_t s = 0;
    for (size_t r = 0; r < n; r++) { const size_t e = s + base + (r < rem ? 1 : 0); out[r] = Slice{s, e}; s = e; }
    return out;
}

// device that owns a device pointer (-1: not a device pointer HIP knows)
int device_of_ptr(const void *ptr) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return at.device;
}

// A shard's view of one batch argument: rows [lo, hi) of `base` ([B][row_bytes]).  Host memory and device memory on the
// shard's own device are used in place; device memory on another device goes through a staging buffer on the shard's
// device, filled / drained by peer copies.
struct ShardArg {
    void *use = nullptr;        // what the shard's context is handed
    void *remote = nullptr;     // the caller's rows when staged
    void *stage = nullptr;
    size_t bytes = 0;
    int dev = 0, remote_dev = 0;
    bool copy_back = false, no_peer = false;
    // device -> device: a peer copy (xGMI where the devices are linked), through a host buffer where that is refused or switched off
    static int across(void *d, int ddev, const void *s, int sdev, size_t n, bool no_peer) {
        if (!no_peer) {
            if (hipMemcpyPeer(d, ddev, s, sdev, n) == hipSuccess) return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : -1;
            (void)hipGetLastError();
        }
        std::vector<unsigned char> bounce(n);
        int cur = 0; (void)hipGetDevice(&cur);
        bool ok = hipSetDevice(sdev) == hipSuccess && hipMemcpy(bounce.data(), s, n, hipMemcpyDeviceToHost) == hipSuccess;
        ok = ok && hipSetDevice(ddev) == hipSuccess && hipMemcpy(d, bounce.data(), n, hipMemcpyHostToDevice) == hipSuccess;
        (void)hipSetDevice(cur);
        return ok ? 0 : -1;
    }
    int prepare(const void *base, size_t row_bytes, size_t lo, size_t hi, int mem, int shard_dev, bool in, bool out, mkt_multi::Stage &pool, bool always, bool nopeer = false) {
        no_peer = nopeer;
        dev = shard_dev;
        bytes = (hi - lo) * row_bytes;
        char *rows = (char *)const_cast<void *>(base) + lo * row_bytes;
        if (mem == MKT_MEM_HOST || !bytes) { use = rows; return 0; }
        const int owner = device_of_ptr(base);
        if (owner < 0 || (owner == shard_dev && !always)) { use = rows; return 0; }
        remote = rows; remote_dev = owner; copy_back = out;
        if (pool.cap < bytes) {
            if (pool.p) (void)hipFree(pool.p);
            pool.p = nullptr; pool.cap = 0;
            if (hipMalloc(&pool.p, bytes) != hipSuccess) return -1;
            pool.cap = bytes;
        }
        stage = pool.p;
        if (in && across(stage, dev, remote, remote_dev, bytes, no_peer) != 0) return -1;   // the shard's own stream is non-blocking: the copy must have landed before its kernels start
        use = stage;
        return 0;
    }
    int finish() {
        int rc = 0;
        if (stage && copy_back && across(remote, remote_dev, stage, dev, bytes, no_peer) != 0) rc = -1;
        stage = nullptr;                       // the buffer stays with the shard (mkt_multi::stage)
        return rc;
    }
};

// run fn(shard, lo, hi) for every non-empty slice, one host thread per shard; first failing shard's status and message win
int run_sharded(mkt_multi *m, size_t B, const std::function<int(int, size_t, size_t, std::string &)> &fn) {
    if (!m->sealed) return mfail(m, MKT_ERR_STATE, "mkt_multi_replicate has not been called");
    const size_t n = m->ctx.size();
    const std::vector<Slice> sl = slices(B, n);
    std::vector<int> rc(n, MKT_OK);
    std::vector<std::string> msg(n);
    auto body = [&](size_t s) {
        if (sl[s].lo == sl[s].hi) return;
        if (hipSetDevice(m->devices[s]) != hipSuccess) { rc[s] = MKT_ERR_HIP; msg[s] = "hipSetDevice failed"; return; }
        rc[s] = fn((int)s, sl[s].lo, sl[s].hi, msg[s]);
    };
    int caller_dev = -1;
    if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;   // shard 0 runs on the calling thread: its current device is put back below
    std::vector<std::thread> th;
    std::vector<size_t> inline_shards;                        // shards whose thread could not be started run on the calling thread
    for (size_t s = 1; s < n; s++) {
        try { th.emplace_back(body, s); } catch (...) { inline_shards.push_back(s); }   // nothing may escape the C ABI
    }
    body(0);                                                  // the calling thread drives shard 0
    for (size_t s : inline_shards) body(s);
    for (auto &t : th) t.join();
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
    for (size_t s = 0; s < n; s++)
        if (rc[s] != MKT_OK) return mfail(m, rc[s], "shard " + std::to_string(s) + " (device " + std::to_string(m->devices[s]) + "): " + msg[s]);
    return MKT_OK;
}

// inputs a caller's stream may still be producing: wait for the owning device (device memory only)
void settle_inputs(const void *ptr, int mem) {
    if (mem != MKT_MEM_DEVICE || !ptr) return;
    const int owner = device_of_ptr(ptr);
    if (owner < 0) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(owner) == hipSuccess) (void)hipDeviceSynchronize();
    if (prev >= 0) (void)hipSetDevice(prev);
}

struct ArgSpec { const void *base; size_t row_bytes; bool in, out; };

// the common shape of a sharded batch call: every argument is [B][row_bytes]; `call` receives the shard's pointers in order
int sharded_call(mkt_multi *m, size_t B, int mem, const std::vector<ArgSpec> &specs,
                 const std::function<int(mkt_ctx *, void **, size_t)> &call) {
    for (const ArgSpec &a : specs) if (a.in) settle_inputs(a.base, mem);
    return run_sharded(m, B, [&](int s, size_t lo, size_t hi, std::string &why) -> int {
        std::vector<ShardArg> args(specs.size());
        std::vector<void *> ptrs(specs.size());
        int rc = MKT_OK;
        if (m->stage[s].size() < specs.size()) m->stage[s].resize(specs.size());     // this shard's thread only
        for (size_t i = 0; i < specs.size(); i++) {
            if (args[i].prepare(specs[i].base, specs[i].row_bytes, lo, hi, mem, m->devices[s], specs[i].in, specs[i].out, m->stage[s][i], m->stage_always, m->no_peer) != 0) { rc = MKT_ERR_HIP; why = "staging a remote device buffer failed"; }
            ptrs[i] = args[i].use;
        }
        if (rc == MKT_OK) {
            rc = call(m->ctx[s], ptrs.data(), hi - lo);
            if (rc != MKT_OK) why = mkt_last_error(m->ctx[s]);
            else if (mkt_synchronize(m->ctx[s]) != MKT_OK) { rc = MKT_ERR_HIP; why = mkt_last_error(m->ctx[s]); }
        }
        for (auto &a : args) if (a.finish() != 0 && rc == MKT_OK) { rc = MKT_ERR_HIP; why = "peer copy of a result slice failed"; }
        return rc;
    });
}

}  // namespace

extern "C" {

const char *mkt_multi_last_error(const mkt_multi *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

int mkt_multi_create(const mkt_params *params, int arith_mode, const int *devices, int nshards, int flags, mkt_multi **out) {
    if (!params || !devices || !out || nshards < 1 || nshards > 1024 || (flags & ~(MKT_MULTI_PRIVATE_KEYS | MKT_MULTI_STAGE_ALWAYS | MKT_MULTI_NO_PEER))) return mfail(nullptr, MKT_ERR_ARG, "bad argument");
    *out = nullptr;
    auto *m = new mkt_multi();
    m->p = *params; m->arith = arith_mode; m->stage_always = (flags & MKT_MULTI_STAGE_ALWAYS) != 0; m->no_peer = (flags & MKT_MULTI_NO_PEER) != 0;
    m->devices.assign(devices, devices + nshards);
    m->ctx.assign((size_t)nshards, nullptr);
    m->root_of.assign((size_t)nshards, -1);
    m->stage.assign((size_t)nshards, {});
    for (int s = 0; s < nshards; s++) {
        // MKT_MULTI_PRIVATE_KEYS: shards that share a device still get their own replicated key copy (the replication path on a one-GPU box)
        if (!(flags & MKT_MULTI_PRIVATE_KEYS)) for (int r = 0; r < s; r++) if (m->devices[r] == m->devices[s]) { m->root_of[s] = m->root_of[r]; break; }
        if (m->root_of[s] >= 0) continue;                     // a logical shard: forked at mkt_multi_replicate
        m->root_of[s] = s;
        const int rc = mkt_ctx_create(params, arith_mode, m->devices[s], &m->ctx[s]);
        if (rc != MKT_OK) { g_multi_create_error = std::string("device ") + std::to_string(m->devices[s]) + ": " + mkt_last_error(nullptr); mkt_multi_destroy(m); return rc; }
    }
    *out = m;
    return MKT_OK;
}

int mkt_multi_destroy(mkt_multi *m) {
    if (!m) return MKT_OK;
    for (size_t s = 0; s < m->stage.size(); s++)
        for (auto &st : m->stage[s]) if (st.p) { int prev = -1; (void)hipGetDevice(&prev); (void)hipSetDevice(m->devices[s]); (void)hipFree(st.p); if (prev >= 0) (void)hipSetDevice(prev); }
    for (size_t s = m->ctx.size(); s-- > 0;) if (m->ctx[s]) (void)mkt_ctx_destroy(m->ctx[s]);   // forks first, roots last (either order is safe: the key set is reference counted)
    delete m;
    return MKT_OK;
}

int mkt_multi_nshards(const mkt_multi *m) { return m ? (int)m->ctx.size() : MKT_ERR_ARG; }
int mkt_multi_device(const mkt_multi *m, int shard) { return (m && shard >= 0 && shard < (int)m->devices.size()) ? m->devices[shard] : MKT_ERR_ARG; }
mkt_ctx *mkt_multi_ctx(mkt_multi *m, int shard) { return (m && shard >= 0 && shard < (int)m->ctx.size()) ? m->ctx[shard] : nullptr; }

int mkt_multi_shard_range(const mkt_multi *m, size_t B, int shard, size_t *lo, size_t *hi) {
    if (!m || !lo || !hi || shard < 0 || shard >= (int)m->ctx.size()) return MKT_ERR_ARG;
    const Slice s = slices(B, m->ctx.size())[(size_t)shard];
    *lo = s.lo; *hi = s.hi;
    return MKT_OK;
}

// ---- keys: uploaded (or generated) once, on the first device ----
#define MKT_MULTI_UNSEALED(m) do { if (!(m)) return MKT_ERR_ARG; if ((m)->sealed) return mfail((m), MKT_ERR_STATE, "the key set is replicated and immutable"); } while (0)
#define MKT_MULTI_FWD(m, call) do { const int _r = (call); if (_r != MKT_OK) return mfail((m), _r, mkt_last_error((m)->ctx[0])); return MKT_OK; } while (0)
int mkt_multi_load_brk(mkt_multi *m, int party, const void *data, int fmt) { MKT_MULTI_UNSEALED(m); MKT_MULTI_FWD(m, mkt_load_brk(m->ctx[0], party, data, fmt)); }
int mkt_multi_load_ksk(mkt_multi *m, int party, const uint32_t *data) { MKT_MULTI_UNSEALED(m); MKT_MULTI_FWD(m, mkt_load_ksk(m->ctx[0], party, data)); }
int mkt_multi_load_rlk(mkt_multi *m, int party, const void *d, const void *f, int fmt) { MKT_MULTI_UNSEALED(m); MKT_MULTI_FWD(m, mkt_load_rlk(m->ctx[0], party, d, f, fmt)); }
int mkt_multi_load_pubkey(mkt_multi *m, int party, const void *b, int fmt) { MKT_MULTI_UNSEALED(m); MKT_MULTI_FWD(m, mkt_load_pubkey(m->ctx[0], party, b, fmt)); }
int mkt_multi_load_crs(mkt_multi *m, const void *a, int fmt) { MKT_MULTI_UNSEALED(m); MKT_MULTI_FWD(m, mkt_load_crs(m->ctx[0], a, fmt)); }
int mkt_multi_keygen_device(mkt_multi *m, int party, const mkt_client_party *keys, const void *crs) { MKT_MULTI_UNSEALED(m); MKT_MULTI_FWD(m, mkt_keygen_device(m->ctx[0], party, keys, crs)); }

// replicate the resident key set of the first device onto every other device (peer copy), fork the logical shards; from here
// on the keys are immutable and the batch entry points may be called
int mkt_multi_replicate(mkt_multi *m) {
    if (!m) return MKT_ERR_ARG;
    if (m->sealed) return MKT_OK;
    const int n = (int)m->ctx.size();
    for (int s = 1; s < n; s++) {
        if (m->root_of[s] != s) continue;
        const int rc = mkt_internal_clone_keys(m->ctx[0], m->ctx[s], m->no_peer ? 1 : 0);
        if (rc != MKT_OK) return mfail(m, rc, std::string("replicating keys to device ") + std::to_string(m->devices[s]) + ": " + mkt_last_error(m->ctx[s]));
    }
    for (int s = 0; s < n; s++) {
        if (m->root_of[s] == s) continue;
        const int rc = mkt_ctx_fork(m->ctx[m->root_of[s]], &m->ctx[s]);
        if (rc != MKT_OK) return mfail(m, rc, std::string("forking a logical shard: ") + mkt_last_error(m->ctx[m->root_of[s]]));
    }
    m->sealed = true;
    return MKT_OK;
}

int mkt_multi_set_option(mkt_multi *m, const char *name, int value) {
    if (!m) return MKT_ERR_ARG;
    for (mkt_ctx *c : m->ctx) if (c) { const int rc = mkt_set_option(c, name, value); if (rc != MKT_OK) return mfail(m, rc, mkt_last_error(c)); }
    return MKT_OK;
}

// ---- batched hot path, sharded ----
int mkt_multi_gate_batch(mkt_multi *m, int op, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int mem) {
    if (!m || !x || !y || !out) return mfail(m, MKT_ERR_ARG, "bad argument");
    const size_t rb = mkt_internal_lwe_len(m->ctx[0]) * 4;
    return sharded_call(m, B, mem, {{x, rb, true, false}, {y, rb, true, false}, {out, rb, false, true}},
                        [&](mkt_ctx *c, void **a, size_t nb) { return mkt_gate_batch(c, op, (const uint32_t *)a[0], (const uint32_t *)a[1], (uint32_t *)a[2], nb, mem); });
}

int mkt_multi_gate_batch_ops(mkt_multi *m, const uint8_t *ops, const uint32_t *x, const uint32_t *y, uint32_t *out, size_t B, int mem) {
    if (!m || !ops || !x || !y || !out) return mfail(m, MKT_ERR_ARG, "bad argument");
    const size_t rb = mkt_internal_lwe_len(m->ctx[0]) * 4;
    return sharded_call(m, B, mem, {{ops, 1, true, false}, {x, rb, true, false}, {y, rb, true, false}, {out, rb, false, true}},
                        [&](mkt_ctx *c, void **a, size_t nb) { return mkt_gate_batch_ops(c, (const uint8_t *)a[0], (const uint32_t *)a[1], (const uint32_t *)a[2], (uint32_t *)a[3], nb, mem); });
}

int mkt_multi_mux_batch(mkt_multi *m, const uint32_t *sel, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t B, int mem) {
    if (!m || !sel || !a || !b || !out) return mfail(m, MKT_ERR_ARG, "bad argument");
    const size_t rb = mkt_internal_lwe_len(m->ctx[0]) * 4;
    return sharded_call(m, B, mem, {{sel, rb, true, false}, {a, rb, true, false}, {b, rb, true, false}, {out, rb, false, true}},
                        [&](mkt_ctx *c, void **p, size_t nb) { return mkt_mux_batch(c, (const uint32_t *)p[0], (const uint32_t *)p[1], (const uint32_t *)p[2], (uint32_t *)p[3], nb, mem); });
}

int mkt_multi_bootstrap_batch(mkt_multi *m, uint32_t *lwe, size_t B, int mem) {
    if (!m || !lwe) return mfail(m, MKT_ERR_ARG, "bad argument");
    const size_t rb = mkt_internal_lwe_len(m->ctx[0]) * 4;
    return sharded_call(m, B, mem, {{lwe, rb, true, true}}, [&](mkt_ctx *c, void **a, size_t nb) { return mkt_bootstrap_batch(c, (uint32_t *)a[0], nb, mem); });
}

int mkt_multi_not_batch(mkt_multi *m, uint32_t *x, size_t B, int mem) {
    if (!m || !x) return mfail(m, MKT_ERR_ARG, "bad argument");
    const size_t rb = mkt_internal_lwe_len(m->ctx[0]) * 4;
    return sharded_call(m, B, mem, {{x, rb, true, true}}, [&](mkt_ctx *c, void **a, size_t nb) { return mkt_not_batch(c, (uint32_t *)a[0], nb, mem); });
}

int mkt_multi_blindrotate_batch(mkt_multi *m, const uint32_t *atilde, void *acc, size_t B, int mem) {
    if (!m || !atilde || !acc) return mfail(m, MKT_ERR_ARG, "bad argument");
    const size_t ra = (mkt_internal_lwe_len(m->ctx[0]) - 1) * 4, rc_ = mkt_internal_acc_bytes(m->ctx[0]);
    return sharded_call(m, B, mem, {{atilde, ra, true, false}, {acc, rc_, true, true}},
                        [&](mkt_ctx *c, void **a, size_t nb) { return mkt_blindrotate_batch(c, (const uint32_t *)a[0], a[1], nb, mem); });
}

int mkt_multi_keyswitch_batch(mkt_multi *m, const void *acc, uint32_t *out, size_t B, int mem) {
    if (!m || !acc || !out) return mfail(m, MKT_ERR_ARG, "bad argument");
    const size_t rc_ = mkt_internal_acc_bytes(m->ctx[0]), rb = mkt_internal_lwe_len(m->ctx[0]) * 4;
    return sharded_call(m, B, mem, {{acc, rc_, true, false}, {out, rb, false, true}},
                        [&](mkt_ctx *c, void **a, size_t nb) { return mkt_keyswitch_batch(c, a[0], (uint32_t *)a[1], nb, mem); });
}

}  // extern "C"
