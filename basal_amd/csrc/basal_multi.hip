// basal_multi.hip -- reads sharded over the GPUs of one node, hit records gathered to one of them with RCCL.
//
// The reference fans batches out to host threads (t_SingleAlign, main.cpp:60-92: every thread takes a batch, aligns it, writes under
// a mutex).  Here a batch fans out over GPUs: every GPU holds the whole 2-bit reference and seed index (sharding the INDEX would change
// visitation order, the per-level cap and threshold tightening), takes a contiguous range of the batch's reads
// -- ranges by read number, which is what feeds myrand -- and its fixed-size hit records travel to GPU 0 in ONE collective per batch:
// ncclGather over xGMI.  32 bytes per read: at the kernel's rate that is ~11 GB/s per GPU, far below a link's 150 GB/s, so the single
// direct gather is all the communication there is.  One process drives all GPUs (ncclCommInitAll + group calls).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "basal_core_priv.h"
#include "basal_internal.h"

using namespace basal;

// [begin, end) of the reads rank `rank` of `world` takes of a batch of n: contiguous, sizes differ by at most one.  Pure arithmetic,
// no GPU: basal_amd/dist.py binds it, so the CPU (gloo) tests and bench.py shard exactly as the native code does.
extern "C" void basal_shard_range(uint64_t n, uint32_t rank, uint32_t world, uint64_t *begin, uint64_t *end) {
    if (world == 0) world = 1;
    const uint64_t base = n / world, extra = n % world;
    const uint64_t b = (uint64_t)rank * base + (rank < extra ? rank : extra);
    if (begin) *begin = b;
    if (end) *end = b + base + (rank < extra ? 1 : 0);
}

// librccl.so is several hundred MB: it is bound when the first multi-GPU object is made, not when libbasal_amd.so is loaded (the
// one-GPU command line never pays for it). A process that already holds an RCCL (PyTorch brings its own) keeps using that one.
namespace {
struct Rccl {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
} g_rccl;
bool load_rccl() {
    if (g_rccl.ok) return true;
    void *h = dlopen(nullptr, RTLD_NOW);  // already in the process?
    if (!h || !dlsym(h, "ncclGather")) {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) { set_error(std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "")); return false; }
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(h, "ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(h, "ncclGroupEnd");
    g_rccl.Gather = (decltype(g_rccl.Gather))dlsym(h, "ncclGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.Gather && g_rccl.GetErrorString;
    if (!g_rccl.ok) set_error("librccl.so lacks ncclCommInitAll / ncclGather / ncclGroupStart / ...");
    return g_rccl.ok;
}
}  // namespace

struct basal_multi {
    int n = 0;
    std::vector<basal_core_t *> cores;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    struct Dev {
        hipStream_t st = nullptr;
        uint8_t *bases = nullptr; size_t cap_bases = 0;
        basal_read *reads = nullptr; size_t cap_reads = 0;
        basal_stale *stales = nullptr; size_t cap_stales = 0;
        uint32_t *order = nullptr; size_t cap_order = 0;
        basal_result *results = nullptr; size_t cap_results = 0;  // own shard (shard_cap records)
        basal_hit *stream = nullptr; size_t cap_stream = 0;
        unsigned long long *used = nullptr;
        unsigned int *counter = nullptr;
    };
    std::vector<Dev> dev;
    std::vector<uint64_t> h2d_bytes;  // per GPU: what the last batch copied to it (basal_multi_last_h2d_bytes)
    // on GPU 0: every rank's records / streams, rank-major
    basal_result *g_results = nullptr; size_t cap_g_results = 0;
    basal_hit *g_stream = nullptr; size_t cap_g_stream = 0;
    unsigned long long *g_used = nullptr;
};

#define HIP_TRYM(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return BASAL_EDEVICE; } } while (0)
#define NCCL_TRYM(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { set_error(std::string(#x) + ": " + g_rccl.GetErrorString(r_)); return BASAL_EDEVICE; } } while (0)

extern "C" void basal_multi_destroy(basal_multi_t *m) {
    if (!m) return;
    for (int d = 0; d < m->n; d++) {
        hipSetDevice(m->devices[(size_t)d]);
        basal_multi::Dev &v = m->dev[(size_t)d];
        hipFree(v.bases); hipFree(v.reads); hipFree(v.stales); hipFree(v.order); hipFree(v.results); hipFree(v.stream); hipFree(v.used); hipFree(v.counter);
        if (v.st) hipStreamDestroy(v.st);
        if ((size_t)d < m->comms.size() && m->comms[(size_t)d] && g_rccl.ok) g_rccl.CommDestroy(m->comms[(size_t)d]);
    }
    if (m->n) { hipSetDevice(m->devices[0]); hipFree(m->g_results); hipFree(m->g_stream); hipFree(m->g_used); }
    for (auto *c : m->cores) basal_core_destroy(c);
    delete m;
}

extern "C" int basal_multi_create(const basal_params *p, const int *devices, int ndev, basal_multi_t **out) {
    if (!p || !devices || ndev < 1 || ndev > 16 || !out) { set_error("multi_create: bad argument (1..16 devices)"); return BASAL_EINVAL; }
    for (int a = 0; a < ndev; a++)
        for (int b = a + 1; b < ndev; b++)
            if (devices[a] == devices[b]) { set_error("multi_create: a GPU is listed twice"); return BASAL_EINVAL; }
    basal_multi *m = new basal_multi();
    m->n = ndev;
    m->devices.assign(devices, devices + ndev);
    m->dev.resize((size_t)ndev);
    for (int d = 0; d < ndev; d++) {
        basal_core_t *c = nullptr;
        int rc = basal_core_create(p, devices[d], &c);
        if (rc) { basal_multi_destroy(m); return rc; }
        m->cores.push_back(c);
    }
    m->comms.assign((size_t)ndev, nullptr);
    {
        if (!load_rccl()) { basal_multi_destroy(m); return BASAL_EDEVICE; }
        ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), ndev, devices);
        if (r != ncclSuccess) { set_error(std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); basal_multi_destroy(m); return BASAL_EDEVICE; }
    }
    for (int d = 0; d < ndev; d++) {
        if (hipSetDevice(devices[d]) != hipSuccess || hipStreamCreateWithFlags(&m->dev[(size_t)d].st, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&m->dev[(size_t)d].used, 8) != hipSuccess || hipMalloc(&m->dev[(size_t)d].counter, 32 * sizeof(unsigned int)) != hipSuccess ||
            hipMemset(m->dev[(size_t)d].counter, 0, 32 * sizeof(unsigned int)) != hipSuccess) {
            set_error("multi_create: stream / counter set-up failed");
            basal_multi_destroy(m);
            return BASAL_EDEVICE;
        }
    }
    *out = m;
    return BASAL_OK;
}

extern "C" int basal_multi_ndev(const basal_multi_t *m) { return m ? m->n : 0; }
extern "C" uint64_t basal_multi_last_h2d_bytes(const basal_multi_t *m, int rank) { return m && rank >= 0 && rank < m->n && (size_t)rank < m->h2d_bytes.size() ? m->h2d_bytes[(size_t)rank] : 0; }
extern "C" basal_core_t *basal_multi_core(basal_multi_t *m, int rank) { return m && rank >= 0 && rank < m->n ? m->cores[(size_t)rank] : nullptr; }

// the whole reference + index on every GPU
extern "C" int basal_multi_upload(basal_multi_t *m, const basal_ref_t *r, int build_index_on_gpu, uint32_t *max_kmer_num) {
    if (!m || !r) { set_error("multi_upload: null argument"); return BASAL_EINVAL; }
    // every GPU stages the reference and builds its index at the same time, one host thread each
    std::vector<int> rcs((size_t)m->n, 0);
    std::vector<uint32_t> mks((size_t)m->n, 0);
    std::vector<std::string> errs((size_t)m->n);
    std::vector<std::thread> th;
    for (int d = 0; d < m->n; d++)
        th.emplace_back([&, d] {
            rcs[(size_t)d] = basal_host_ref_upload(r, m->cores[(size_t)d], build_index_on_gpu, &mks[(size_t)d]);
            if (rcs[(size_t)d]) errs[(size_t)d] = basal_last_error();  // (the message is per thread)
        });
    for (auto &t : th) t.join();
    for (int d = 0; d < m->n; d++)
        if (rcs[(size_t)d]) { set_error(errs[(size_t)d]); return rcs[(size_t)d]; }
    if (max_kmer_num) *max_kmer_num = mks[0];
    return BASAL_OK;
}

template <typename T>
static int grow_dev(T *&p, size_t &cap, size_t need) {
    if (need <= cap) return BASAL_OK;
    hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t nc = need + need / 4 + 1024;
    HIP_TRYM(hipMalloc(&p, nc * sizeof(T)));
    cap = nc;
    return BASAL_OK;
}

// basal_core_align_batch over all GPUs: same arguments, same results. Every GPU receives the bases, descriptors and stale entries of its own
// range of reads (and of the reads those inherit a start offset from, which may lie in another shard), aligns that range, and GPU 0 gathers the records.
extern "C" int basal_multi_align_batch(basal_multi_t *m, const uint8_t *bases, uint64_t nbases, const basal_read *reads, uint32_t n, const basal_stale *stales,
                                       uint32_t nstale, int stream_mode, basal_result *results, basal_hit *stream, uint64_t stream_cap, uint64_t *stream_used,
                                       uint8_t carry[2][2]) {
    if (!m || (n && (!bases || !reads || !results))) { set_error("multi_align_batch: null argument"); return BASAL_EINVAL; }
    if (stream_mode != BASAL_STREAM_NONE && (!stream || !stream_used)) { set_error("multi_align_batch: stream buffers required for this stream_mode"); return BASAL_EINVAL; }
    if (stream_used) *stream_used = 0;
    if (n == 0) return BASAL_OK;
    const int N = m->n;
    uint32_t max_len = 0;
    if (int vrc = basal_validate_batch(m->cores[0]->p, reads, n, nbases, stales, nstale, "multi_align_batch", &max_len)) return vrc;
    if (max_len == 0) max_len = 1;
    const uint64_t shard_cap = (n + (uint64_t)N - 1) / (uint64_t)N;           // records every rank sends (the last ranks' tails are padding)
    const uint64_t stream_cap_dev = stream_mode == BASAL_STREAM_NONE ? 0 : (stream_cap + (uint64_t)N - 1) / (uint64_t)N + 1024;
    std::vector<uint32_t> iota(n);
    for (uint32_t i = 0; i < n; i++) iota[i] = i;
    m->h2d_bytes.assign((size_t)N, 0);
    // 1. inputs to every GPU, each GPU's launch on its own stream
    for (int d = 0; d < N; d++) {
        basal_multi::Dev &v = m->dev[(size_t)d];
        basal_core *c = m->cores[(size_t)d];
        HIP_TRYM(hipSetDevice(m->devices[(size_t)d]));
        int rc;
        if ((rc = grow_dev(v.bases, v.cap_bases, nbases + 64)) || (rc = grow_dev(v.reads, v.cap_reads, (size_t)n)) || (rc = grow_dev(v.order, v.cap_order, (size_t)shard_cap)) ||
            (rc = grow_dev(v.results, v.cap_results, (size_t)n)) || (nstale && (rc = grow_dev(v.stales, v.cap_stales, (size_t)nstale))) ||
            (stream_cap_dev && (rc = grow_dev(v.stream, v.cap_stream, (size_t)stream_cap_dev)))) return rc;
        uint64_t lo, hi;
        basal_shard_range(n, (uint32_t)d, (uint32_t)N, &lo, &hi);
        // H2D per GPU = what its shard needs, at the offsets the whole batch would have: the bases, descriptors and stale entries of its own
        // reads and of the earlier reads they inherit a start offset from (basal_stale.src may name a read of another shard)
        if (hi > lo) {
            uint64_t b0 = ~0ull, b1 = 0, d0 = lo, s0 = ~0ull, s1 = 0;
            auto need = [&](uint32_t i) {
                const basal_read &r = reads[i];
                if (r.len == 0) return;
                b0 = std::min<uint64_t>(b0, r.seq_off);
                b1 = std::max<uint64_t>(b1, (uint64_t)r.seq_off + r.len);
            };
            for (uint64_t i = lo; i < hi; i++) {
                need((uint32_t)i);
                const basal_read &r = reads[i];
                if (r.len && r.stale_idx != BASAL_STALE_NONE && stales) {
                    s0 = std::min<uint64_t>(s0, r.stale_idx);
                    s1 = std::max<uint64_t>(s1, (uint64_t)r.stale_idx + 1);
                    const uint32_t src = stales[r.stale_idx].src;
                    if (src != BASAL_STALE_CARRY && src < i) { need(src); d0 = std::min<uint64_t>(d0, src); }
                }
            }
            if (b1 > b0) HIP_TRYM(hipMemcpyAsync(v.bases + b0, bases + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, v.st));
            HIP_TRYM(hipMemcpyAsync(v.reads + d0, reads + d0, (size_t)(hi - d0) * sizeof(basal_read), hipMemcpyHostToDevice, v.st));
            if (s1 > s0) HIP_TRYM(hipMemcpyAsync(v.stales + s0, stales + s0, (size_t)(s1 - s0) * sizeof(basal_stale), hipMemcpyHostToDevice, v.st));
            m->h2d_bytes[(size_t)d] = (b1 > b0 ? b1 - b0 : 0) + (hi - d0) * sizeof(basal_read) + (s1 > s0 ? (s1 - s0) * sizeof(basal_stale) : 0);
        } else m->h2d_bytes[(size_t)d] = 0;
        if (hi > lo) HIP_TRYM(hipMemcpyAsync(v.order, iota.data() + lo, (size_t)(hi - lo) * 4, hipMemcpyHostToDevice, v.st));
        HIP_TRYM(hipMemsetAsync(v.used, 0, 8, v.st));
        HIP_TRYM(hipMemsetAsync(v.results + lo, 0, (size_t)shard_cap * sizeof(basal_result) <= (v.cap_results - lo) * sizeof(basal_result) ? (size_t)shard_cap * sizeof(basal_result) : (size_t)(hi - lo) * sizeof(basal_result), v.st));
        if (hi > lo) {
            basal_align_extra ex;
            ex.order = v.order;
            ex.counter = v.counter;
            // (carry: the start offset inherited from the previous batch is the same on every GPU)
            rc = basal_launch_align_carry(c, v.bases, nbases, v.reads, (uint32_t)(hi - lo), nstale ? v.stales : nullptr, nstale, max_len, stream_mode, v.results, v.stream,
                                          stream_cap_dev, v.used, carry, v.st, &ex);
            if (rc) {  // (the GPUs already started still read `iota` and the caller's buffers: let them finish before those go away)
                for (int e = 0; e <= d; e++) { hipSetDevice(m->devices[(size_t)e]); hipStreamSynchronize(m->dev[(size_t)e].st); }
                return rc;
            }
        }
    }
    // 2. the one collective: every GPU's shard of records (and hit-stream records) to GPU 0
    HIP_TRYM(hipSetDevice(m->devices[0]));
    {
        int rc;
        if ((rc = grow_dev(m->g_results, m->cap_g_results, (size_t)(shard_cap * (uint64_t)N)))) return rc;
        if (stream_cap_dev && (rc = grow_dev(m->g_stream, m->cap_g_stream, (size_t)(stream_cap_dev * (uint64_t)N)))) return rc;
        if (!m->g_used) HIP_TRYM(hipMalloc(&m->g_used, 8 * 16));
    }
    NCCL_TRYM(g_rccl.GroupStart());
    for (int d = 0; d < N; d++) {
        basal_multi::Dev &v = m->dev[(size_t)d];
        uint64_t lo, hi;
        basal_shard_range(n, (uint32_t)d, (uint32_t)N, &lo, &hi);
        NCCL_TRYM(g_rccl.Gather(v.results + lo, m->g_results, (size_t)shard_cap * sizeof(basal_result), ncclUint8, 0, m->comms[(size_t)d], v.st));
    }
    NCCL_TRYM(g_rccl.GroupEnd());
    if (stream_cap_dev) {
        NCCL_TRYM(g_rccl.GroupStart());
        for (int d = 0; d < N; d++) NCCL_TRYM(g_rccl.Gather(m->dev[(size_t)d].stream, m->g_stream, (size_t)stream_cap_dev * sizeof(basal_hit), ncclUint8, 0, m->comms[(size_t)d], m->dev[(size_t)d].st));
        NCCL_TRYM(g_rccl.GroupEnd());
        NCCL_TRYM(g_rccl.GroupStart());
        for (int d = 0; d < N; d++) NCCL_TRYM(g_rccl.Gather(m->dev[(size_t)d].used, m->g_used, 8, ncclUint8, 0, m->comms[(size_t)d], m->dev[(size_t)d].st));
        NCCL_TRYM(g_rccl.GroupEnd());
    }
    // 3. GPU 0 -> host, shard by shard (the padding behind short shards is skipped); ledgers of every GPU
    std::vector<unsigned long long> used((size_t)N, 0);
    for (int d = 0; d < N; d++) {
        uint64_t lo, hi;
        basal_shard_range(n, (uint32_t)d, (uint32_t)N, &lo, &hi);
        if (hi > lo) HIP_TRYM(hipMemcpyAsync(results + lo, m->g_results + (size_t)d * shard_cap, (size_t)(hi - lo) * sizeof(basal_result), hipMemcpyDeviceToHost, m->dev[0].st));
    }
    if (stream_cap_dev) HIP_TRYM(hipMemcpyAsync(used.data(), m->g_used, 8 * (size_t)N, hipMemcpyDeviceToHost, m->dev[0].st));
    int ret = BASAL_OK;
    for (int d = 0; d < N; d++) {
        HIP_TRYM(hipSetDevice(m->devices[(size_t)d]));
        unsigned int guard[24];
        HIP_TRYM(hipMemcpyAsync(guard, m->dev[(size_t)d].counter + 1, sizeof guard, hipMemcpyDeviceToHost, m->dev[(size_t)d].st));
        HIP_TRYM(hipMemsetAsync(m->dev[(size_t)d].counter + 1, 0, sizeof guard, m->dev[(size_t)d].st));
        HIP_TRYM(hipStreamSynchronize(m->dev[(size_t)d].st));
        if (int gr = basal_report_guard(guard)) ret = gr;
    }
    if (ret) return ret;
    // 4. hit streams: rank-local offsets -> offsets into the caller's one stream
    if (stream_cap_dev) {
        HIP_TRYM(hipSetDevice(m->devices[0]));
        uint64_t at = 0;
        bool overflow = false;
        for (int d = 0; d < N; d++) {
            if (used[(size_t)d] > stream_cap_dev) overflow = true;
            const uint64_t take = used[(size_t)d] < stream_cap_dev ? used[(size_t)d] : stream_cap_dev;
            uint64_t lo, hi;
            basal_shard_range(n, (uint32_t)d, (uint32_t)N, &lo, &hi);
            if (at + take <= stream_cap) {
                if (take) HIP_TRYM(hipMemcpy(stream + at, m->g_stream + (size_t)d * stream_cap_dev, (size_t)take * sizeof(basal_hit), hipMemcpyDeviceToHost));
                for (uint64_t i = lo; i < hi; i++)
                    if (results[i].stream_n && results[i].status != BASAL_READ_OVERFLOW) results[i].stream_first += (uint32_t)at;
            } else overflow = true;
            at += used[(size_t)d];
        }
        // The caller's capacity is split evenly over the GPUs, so what it must offer next time is N x the LARGEST shard's need (a retry sized
        // from the total would starve the fullest shard again, forever if the shards are uneven enough).
        unsigned long long worst = 0;
        for (int d = 0; d < N; d++) worst = std::max(worst, used[(size_t)d]);
        *stream_used = overflow ? (uint64_t)N * (worst + 1024) : at;
        if (overflow) { set_error("multi_align_batch: hit stream too small; a capacity of " + std::to_string(*stream_used) + " records fits every GPU's share"); ret = BASAL_EOVERFLOW; }
    }
    // carry: as basal_core_align_batch leaves it (the start offset after the last aligned read of each slot)
    if (carry) {
        const basal_params &P = m->cores[0]->p;
        for (int slot = 0; slot < 2; slot++)
            for (uint32_t i = n; i-- > 0;) {
                const basal_read &r = reads[i];
                const uint32_t rs = r.readset & 0x7fu;
                if (r.len == 0 || (rs == 2 ? 1 : 0) != slot || results[i].status == BASAL_READ_SKIPPED) continue;
                const bool f0 = (P.chains == 1) || ((P.chains <= 1) == (rs < 2)), f1 = (P.chains == 1) || ((P.chains <= 1) == (rs == 2));
                if (f0) carry[slot][0] = results[i].start_off[0];
                if (f1) carry[slot][1] = results[i].start_off[1];
                break;
            }
    }
    return ret;
}
