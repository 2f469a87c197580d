// basal_pipe.hip -- the batch pipeline around the align kernel: raw read text (or raw read tables, or prepared reads) in
// page-locked host buffers -> HBM -> device-side parse / FilterReads / inherited-state table (basal_prep.hip) -> align kernels (one
// launch per read-length class) -> device-side SAM text -> page-locked host buffer.  Three HIP streams, one per resource: copies in
// (PCIe down), kernels, copies out (PCIe up); a batch moves from one to the next through events, so the H2D copy of batch k+1 and the
// D2H copy of batch k-1 overlap the kernels of batch k while each resource serves the batches in order at its full rate (batches
// sharing a resource -- two persistent align grids on the GPU, three copies on the link -- only finish later, all of them).  The
// kernel stream also orders the small carry state (what later reads inherit from earlier ones).  Nothing is decided on the host from
// device data until a batch is collected: batch sizes the GPU counted itself (text form) stay on the GPU.
//
// Replaces, for the `basal` command line, the reference's per-thread loop LoadBatchReads -> ImportBatchReads -> Do_Batch ->
// write _str_align (main.cpp:60-92).
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "basal_bits.h"
#include "basal_internal.h"
#include "basal_prep.h"

using namespace basal;

namespace {

#define HIP_TRYQ(x)                                                    \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            set_error(std::string(#x) + ": " + hipGetErrorString(e_)); \
            return BASAL_EDEVICE;                                      \
        }                                                              \
    } while (0)

enum { EV_START = 0, EV_H2D, EV_COMP0, EV_PREP, EV_ALIGN, EV_FORMAT, EV_COUNTERS, EV_OUT0, EV_OUT, EV_N };
enum { MODE_TEXT = 0, MODE_RECORDS, MODE_PREPARED };
enum { ST_FREE = 0, ST_ACQUIRED, ST_INFLIGHT, ST_HELD };

struct Slot {
    SlotDev d;
    int dev = 0;  // index into basal_pipe::devs: the GPU this slot's buffers live on
    uint8_t *h_blob = nullptr;
    basal_rawread *h_raw = nullptr;
    uint8_t *h_out = nullptr;
    size_t h_out_cap = 0;
    BatchCounters *h_cnt = nullptr;
    unsigned int *h_guard = nullptr;
    hipEvent_t ev[EV_N] = {nullptr};
    int state = ST_FREE;
    uint32_t batch_no = 0;
    int mode = MODE_TEXT;
    uint32_t n_host = 0, max_len = 0;
    uint64_t nbytes = 0;
    hipStream_t comp = nullptr;  // the kernel stream of the batch this slot holds (one of its GPU's two, by batch number)
    bool out_queued = false;  // the D2H copy of the output has been queued
    bool collecting = false;  // basal_pipe_collect is working on this slot (the submitter leaves it alone)
    uint64_t out_bytes = 0;
};

}  // namespace

// one GPU of a pipe: its core, its ring of carry states and its four streams
struct PipeDev {
    basal_core *c = nullptr;
    PrepShared sh;
    hipStream_t st_in = nullptr, st_comp = nullptr, st_out = nullptr, st_cnt = nullptr;  // st_cnt: the few bytes of counters per batch (the host sizes the output copy from them; they must not queue behind an output copy)
    // A second kernel stream: consecutive batches' kernels alternate between the two, so the END of one batch's align launch -- a few waves finishing
    // its longest reads, 10-25 ms on a repeat-rich index whatever the batch size -- overlaps the next batch's kernels instead of idling the GPU.
    // What orders the batches is the carry state alone: batch b + 1's prep kernels wait for the event behind batch b's (EV_PREP), as they do across GPUs.
    // (The two streams must land on different hardware queues to overlap: HIP maps streams of one priority onto few queues -- the `basal` command line
    // sets GPU_MAX_HW_QUEUES=8 before HIP starts; a host that embeds the library should do the same.)
    hipStream_t st_comp2 = nullptr;
};

// A pipe over one GPU, or over several (basal_pipe_create_multi): whole batches fan out over the GPUs, the way the reference's worker
// threads each take a whole batch (main.cpp:60-92), every batch staying on one GPU from its text to its SAM bytes -- H2D bytes per read do
// not grow with the number of GPUs, and no GPU waits for another except for the carry state: what a SingleAlign object carries from read
// to read (CarryState, 470 KB) travels from the GPU that prepared batch b-1 to the one that prepares batch b, behind b-1's prep kernels
// (the align kernels, 90 % of the work, overlap freely). One sequence of batch numbers, one order of results: the output is the
// one-GPU output, byte for byte.
struct basal_pipe {
    std::vector<PipeDev> devs;
    basal_core *c = nullptr;  // devs[0].c (parameters, limits)
    basal_pipe_opts o;
    PrepConst k;
    uint32_t ncarry = 0;            // carry states per GPU, indexed by batch number: batch b reads [b % ncarry] and writes [(b + 1) % ncarry]
    std::vector<int> carry_dev;     // [ncarry]: the GPU that holds the valid copy of that state (-1: every GPU does -- the initial state)
    std::vector<int> carry_slot;    // [ncarry]: the slot whose prep kernels wrote it (-1: none pending), for the event a reader on another GPU waits for
    std::vector<Slot> slots;
    std::mutex m;
    std::condition_variable cv;
    int acquired = -1, held = -1, last_slot = -1;
    std::deque<int> inflight;
    uint32_t next_batch = 0;
    uint32_t read_end = 0xFFFFFFFFu;
    bool broken = false;     // a batch was refused (or basal_pipe_stop): no acquire / submit / collect until basal_pipe_rewind
    uint32_t rewind_to = 0;  // the batch number the pipe continues from after a rewind
};

extern "C" int basal_core_set_contig_names(basal_core_t *c, const char *const *names, uint32_t ncontig) {
    if (!c || !names || ncontig == 0) { set_error("set_contig_names: bad argument"); return BASAL_EINVAL; }
    if (c->have_ref && ncontig != c->ncontig) { set_error("set_contig_names: contig count differs from the staged reference"); return BASAL_EINVAL; }
    HIP_TRYQ(hipSetDevice(c->device));
    std::string blob;
    std::vector<uint32_t> off(ncontig + 1);
    for (uint32_t i = 0; i < ncontig; i++) {
        off[i] = (uint32_t)blob.size();
        blob += names[i] ? names[i] : "*";
    }
    off[ncontig] = (uint32_t)blob.size();
    hipFree(c->d_names); hipFree(c->d_name_off);
    c->d_names = nullptr; c->d_name_off = nullptr;
    HIP_TRYQ(hipMalloc(&c->d_names, blob.size() + 16));
    HIP_TRYQ(hipMalloc(&c->d_name_off, (size_t)(ncontig + 1) * 4));
    HIP_TRYQ(hipMemcpy(c->d_names, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIP_TRYQ(hipMemcpy(c->d_name_off, off.data(), (size_t)(ncontig + 1) * 4, hipMemcpyHostToDevice));
    c->n_names = ncontig;
    return BASAL_OK;
}

static void free_slot(Slot &s) {
    hipFree(s.d.text); hipFree(s.d.raw); hipFree(s.d.desc); hipFree(s.d.aux); hipFree(s.d.stales); hipFree(s.d.npos); hipFree(s.d.bmax1); hipFree(s.d.bmax2);
    hipFree(s.d.defidx); hipFree(s.d.order); hipFree(s.d.results); hipFree(s.d.stream); hipFree(s.d.nl); hipFree(s.d.blk_cnt); hipFree(s.d.out_off); hipFree(s.d.out);
    hipFree(s.d.cnt); hipFree(s.d.counter); hipFree(s.d.scratch); hipFree(s.d.cub_tmp); hipFree(s.d.pe_pairs); hipFree(s.d.pe_recs); hipFree(s.d.pe_work);
    if (s.h_blob) hipHostFree(s.h_blob);
    if (s.h_raw) hipHostFree(s.h_raw);
    if (s.h_out) hipHostFree(s.h_out);
    if (s.h_cnt) hipHostFree(s.h_cnt);
    if (s.h_guard) hipHostFree(s.h_guard);
    for (int i = 0; i < EV_N; i++) if (s.ev[i]) hipEventDestroy(s.ev[i]);
    s = Slot();
}

extern "C" void basal_pipe_destroy(basal_pipe_t *p) {
    if (!p) return;
    for (auto &v : p->devs) { hipSetDevice(v.c->device); hipDeviceSynchronize(); }
    for (auto &s : p->slots) { hipSetDevice(p->devs[(size_t)s.dev].c->device); free_slot(s); }
    for (auto &v : p->devs) {
        hipSetDevice(v.c->device);
        for (uint32_t i = 0; i < v.sh.ncarry; i++) hipFree(v.sh.carry[i]);
        if (v.st_in) hipStreamDestroy(v.st_in);
        if (v.st_comp) hipStreamDestroy(v.st_comp);
        if (v.st_comp2) hipStreamDestroy(v.st_comp2);
        if (v.st_out) hipStreamDestroy(v.st_out);
        if (v.st_cnt) hipStreamDestroy(v.st_cnt);
    }
    delete p;
}

extern "C" int basal_pipe_create(basal_core_t *c, const basal_pipe_opts *o, basal_pipe_t **out) { return basal_pipe_create_multi(&c, 1, o, out); }

extern "C" int basal_pipe_create_multi(basal_core_t *const *cores, int ncores, const basal_pipe_opts *o, basal_pipe_t **out) {
    if (!cores || ncores < 1 || ncores > 16 || !out) { set_error("pipe_create: null argument, or not 1..16 cores"); return BASAL_EINVAL; }
    for (int g = 0; g < ncores; g++) if (!cores[g]) { set_error("pipe_create: null core"); return BASAL_EINVAL; }
    basal_core *c = cores[0];
    basal_pipe_opts op;
    memset(&op, 0, sizeof op);
    if (o) op = *o;
    if (op.depth == 0) op.depth = 3;
    if (op.depth < 2 || op.depth > 8) { set_error("pipe_create: depth must be 2..8"); return BASAL_EINVAL; }
    if (op.max_reads == 0) op.max_reads = 4u << 20;
    op.max_reads = (op.max_reads + 4095u) & ~4095u;  // the group maxima of basal_prep.hip work on 4096 reads
    if (op.max_bytes == 0) op.max_bytes = 512ull << 20;
    op.max_bytes = (op.max_bytes + 4095ull) & ~4095ull;
    if (op.max_bytes >= 0xFFFFF000ull) { set_error("pipe_create: max_bytes must stay below 4 GiB (32-bit offsets inside a batch)"); return BASAL_EINVAL; }
    if (op.output > BASAL_PIPE_OUT_RESULTS) { set_error("pipe_create: bad output mode"); return BASAL_EINVAL; }
    if (op.flags & ~BASAL_PIPE_PAIRS) { set_error("pipe_create: unknown flag"); return BASAL_EINVAL; }
    if ((op.flags & BASAL_PIPE_PAIRS) && op.output != BASAL_PIPE_OUT_SAM) { set_error("pipe_create: BASAL_PIPE_PAIRS goes with BASAL_PIPE_OUT_SAM"); return BASAL_EINVAL; }
    basal_pipe *p = new basal_pipe();
    p->c = c;
    p->o = op;
    prep_make_const(c->p, p->k);
    p->devs.resize((size_t)ncores);
    const uint32_t nslots = op.depth * (uint32_t)ncores;  // `depth` batches in flight per GPU
    p->ncarry = nslots + 1;
    if (p->ncarry > sizeof(p->devs[0].sh.carry) / sizeof(p->devs[0].sh.carry[0])) { basal_pipe_destroy(p); set_error("pipe_create: depth x GPUs must stay below 128 batches in flight"); return BASAL_EINVAL; }
    p->carry_dev.assign(p->ncarry, -1);
    p->carry_slot.assign(p->ncarry, -1);
    p->slots.resize(nslots);
#define TRYD(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); basal_pipe_destroy(p); return e_ == hipErrorOutOfMemory ? BASAL_ENOMEM : BASAL_EDEVICE; } } while (0)
    for (int g = 0; g < ncores; g++) {
        PipeDev &v = p->devs[(size_t)g];
        v.c = cores[g];
        TRYD(hipSetDevice(v.c->device));
        int rc = basal_ensure_launch_geometry(v.c);
        if (rc) { basal_pipe_destroy(p); return rc; }
        v.sh.ncarry = p->ncarry;
        for (uint32_t i = 0; i < p->ncarry; i++) {
            TRYD(hipMalloc(&v.sh.carry[i], sizeof(CarryState)));
            TRYD(hipMemset(v.sh.carry[i], 0, sizeof(CarryState)));
        }
    }
    const uint32_t mr = op.max_reads;
    const size_t cub = prep_cub_tmp_bytes(mr, op.max_bytes);
    const bool sam = op.output == BASAL_PIPE_OUT_SAM;
    const unsigned host_flags = ncores > 1 ? hipHostMallocPortable : hipHostMallocDefault;  // (page-locked for every GPU's copy engines)
    for (uint32_t si = 0; si < nslots; si++) {
        Slot &s = p->slots[si];
        s.dev = (int)(si % (uint32_t)ncores);  // consecutive slots on consecutive GPUs: consecutive batches fan out
        basal_core *cg = p->devs[(size_t)s.dev].c;
        TRYD(hipSetDevice(cg->device));
        SlotDev &d = s.d;
        d.text_cap = op.max_bytes;
        TRYD(hipMalloc(&d.text, op.max_bytes + 1024));
        TRYD(hipMalloc(&d.raw, (size_t)mr * sizeof(basal_rawread)));
        TRYD(hipMalloc(&d.desc, ((size_t)mr + 2) * sizeof(basal_read)));
        TRYD(hipMalloc(&d.results, (size_t)mr * sizeof(basal_result)));
        TRYD(hipMalloc(&d.cnt, sizeof(BatchCounters)));
        TRYD(hipMalloc(&d.counter, 32 * sizeof(unsigned int)));
        TRYD(hipMemset(d.counter, 0, 32 * sizeof(unsigned int)));
        TRYD(hipMalloc(&d.scratch, (size_t)cg->grid * 4 * cg->scratch_per_wave * sizeof(basal_hit)));
        if (sam) {
            TRYD(hipMalloc(&d.aux, (size_t)mr * sizeof(ReadAux)));
            TRYD(hipMalloc(&d.stales, (size_t)mr * sizeof(basal_stale)));
            TRYD(hipMalloc(&d.npos, (size_t)6 * mr * sizeof(uint16_t)));
            TRYD(hipMalloc(&d.bmax1, (size_t)2 * (mr / 64) * sizeof(uint16_t)));
            TRYD(hipMalloc(&d.bmax2, (size_t)2 * (mr / 4096) * sizeof(uint16_t)));
            TRYD(hipMalloc(&d.defidx, (size_t)4 * mr * sizeof(int32_t)));
            TRYD(hipMalloc(&d.order, ((size_t)3 * mr + 2 * kStackMax + 2) * sizeof(uint32_t)));
            TRYD(hipMalloc(&d.nl, (size_t)4 * mr * sizeof(uint32_t)));
            TRYD(hipMalloc(&d.blk_cnt, (size_t)(op.max_bytes / 4096 + 2) * sizeof(uint32_t)));
            TRYD(hipMalloc(&d.out_off, ((size_t)mr + 1) * sizeof(unsigned long long)));
            d.out_cap = op.max_bytes + op.max_bytes / 2 + (size_t)mr * 64 + 4096;
            TRYD(hipMalloc(&d.out, d.out_cap));
            TRYD(hipMalloc(&d.cub_tmp, cub));
            d.cub_tmp_bytes = cub;
            if (c->p.report_repeat_hits == 2 || (op.flags & BASAL_PIPE_PAIRS)) {  // (pairs: every mate's whole hit log, for the pairing kernel)
                d.stream_cap = (uint64_t)mr * 8 + 4096;
                TRYD(hipMalloc(&d.stream, d.stream_cap * sizeof(basal_hit)));
            }
            if (op.flags & BASAL_PIPE_PAIRS) {
                d.pe_work_cap = d.stream_cap;
                d.pe_recs_cap = (uint64_t)mr + 4096;
                TRYD(hipMalloc(&d.pe_work, d.pe_work_cap * sizeof(basal_hit)));
                TRYD(hipMalloc(&d.pe_recs, d.pe_recs_cap * sizeof(basal_pe_rec)));
                TRYD(hipMalloc(&d.pe_pairs, ((size_t)mr / 2 + 1) * sizeof(basal_pe_pair)));
            }
            s.h_out_cap = d.out_cap;
        } else s.h_out_cap = (size_t)mr * sizeof(basal_result);
        TRYD(hipHostMalloc(&s.h_blob, op.max_bytes, host_flags));
        TRYD(hipHostMalloc(&s.h_raw, (size_t)mr * sizeof(basal_rawread), host_flags));
        TRYD(hipHostMalloc(&s.h_out, s.h_out_cap, host_flags));
        TRYD(hipHostMalloc(&s.h_cnt, sizeof(BatchCounters), host_flags));
        TRYD(hipHostMalloc(&s.h_guard, 24 * sizeof(unsigned int), host_flags));
        for (int i = 0; i < EV_N; i++) TRYD(hipEventCreate(&s.ev[i]));
    }
    for (auto &v : p->devs) {
        // HIP multiplexes streams onto a few hardware queues (4 by default), and two streams on one queue run in submission order --
        // the copy-in stream and the kernel stream sharing a queue would undo the whole pipeline (measured: exactly that happened).
        // Streams of different priority get queues of their own, so the four streams are spread over the priority levels.
        TRYD(hipSetDevice(v.c->device));
        int lo = 0, hi = 0;
        TRYD(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least urgent (numerically largest), hi = most urgent
        const char *e = getenv("BASAL_PIPE_PRIO");
        int pin = hi, pcomp = (lo + hi) / 2, pout = lo, pcnt = hi;
        if (e && strlen(e) == 4) { auto lv = [&](char ch) { return ch == 'h' ? hi : ch == 'l' ? lo : (lo + hi) / 2; }; pin = lv(e[0]); pcomp = lv(e[1]); pout = lv(e[2]); pcnt = lv(e[3]); }
        TRYD(hipStreamCreateWithPriority(&v.st_in, hipStreamNonBlocking, pin));
        TRYD(hipStreamCreateWithPriority(&v.st_comp, hipStreamNonBlocking, pcomp));
        TRYD(hipStreamCreateWithPriority(&v.st_comp2, hipStreamNonBlocking, pcomp));
        TRYD(hipStreamCreateWithPriority(&v.st_out, hipStreamNonBlocking, pout));
        TRYD(hipStreamCreateWithPriority(&v.st_cnt, hipStreamNonBlocking, pcnt));
    }
#undef TRYD
    *out = p;
    return BASAL_OK;
}

extern "C" int basal_pipe_acquire(basal_pipe_t *p, uint8_t **blob, basal_rawread **raw) {
    if (!p) { set_error("pipe_acquire: null argument"); return BASAL_EINVAL; }
    std::unique_lock<std::mutex> lk(p->m);
    if (p->acquired >= 0) { set_error("pipe_acquire: the acquired slot has not been submitted"); return BASAL_ESTATE; }
    for (;;) {
        if (p->broken) { set_error("pipe_acquire: the pipe is stopped (a batch was refused): call basal_pipe_rewind"); return BASAL_ESTATE; }
        for (size_t k = 1; k <= p->slots.size(); k++) {  // (from the slot behind the last one taken: consecutive batches go to consecutive GPUs)
            const size_t i = ((size_t)(p->last_slot + 1) + k - 1) % p->slots.size();
            if (p->slots[i].state == ST_FREE) {
                p->slots[i].state = ST_ACQUIRED;
                p->acquired = (int)i;
                p->last_slot = (int)i;
                if (blob) *blob = p->slots[i].h_blob;
                if (raw) *raw = p->slots[i].h_raw;
                return BASAL_OK;
            }
        }
        // every slot is in flight or held by the collector: wait for a collect to release one
        p->cv.wait(lk);
    }
}

// queue the D2H copy of a finished batch's output if its size is known (its counters have arrived) and it fits
static int try_queue_output(basal_pipe *p, Slot &s, bool wait) {
    if (s.out_queued) return BASAL_OK;
    PipeDev &v = p->devs[(size_t)s.dev];
    HIP_TRYQ(hipSetDevice(v.c->device));
    if (wait) HIP_TRYQ(hipEventSynchronize(s.ev[EV_COUNTERS]));
    else if (hipEventQuery(s.ev[EV_COUNTERS]) != hipSuccess) return BASAL_OK;
    if (p->o.output == BASAL_PIPE_OUT_SAM) {
        const BatchCounters &cn = *s.h_cnt;
        if (cn.irregular || cn.out_bytes > s.d.out_cap) return BASAL_OK;  // collect deals with it
        if (cn.out_bytes > s.h_out_cap) {
            if (!wait) return BASAL_OK;
            hipHostFree(s.h_out);
            s.h_out = nullptr;
            s.h_out_cap = cn.out_bytes + cn.out_bytes / 4;
            HIP_TRYQ(hipHostMalloc(&s.h_out, s.h_out_cap, p->devs.size() > 1 ? hipHostMallocPortable : hipHostMallocDefault));
        }
        s.out_bytes = cn.out_bytes;
        HIP_TRYQ(hipEventRecord(s.ev[EV_OUT0], v.st_out));
        if (s.out_bytes) HIP_TRYQ(hipMemcpyAsync(s.h_out, s.d.out, s.out_bytes, hipMemcpyDeviceToHost, v.st_out));
    }
    HIP_TRYQ(hipEventRecord(s.ev[EV_OUT], v.st_out));
    s.out_queued = true;
    return BASAL_OK;
}

static int queue_align(basal_pipe *p, Slot &s) {
    PipeDev &v = p->devs[(size_t)s.dev];
    basal_core *c = v.c;
    SlotDev &d = s.d;
    const uint32_t mr = p->o.max_reads;
    const int smode = (p->o.flags & BASAL_PIPE_PAIRS) ? BASAL_STREAM_ALL
                      : p->o.output == BASAL_PIPE_OUT_SAM && c->p.report_repeat_hits == 2 ? BASAL_STREAM_BEST : BASAL_STREAM_NONE;
    if (s.mode == MODE_PREPARED) {
        basal_align_extra ex;
        ex.counter = d.counter;
        ex.scratch = d.scratch;
        return basal_launch_align(c, d.text, s.nbytes, d.raw, s.n_host, nullptr, 0, s.max_len, BASAL_STREAM_NONE, d.results, nullptr, 0, &d.cnt->stream_used, s.comp, &ex);
    }
    static const uint32_t cls_len[3] = {128, 256, BASAL_MAXREADLEN};
    for (int cl = 0; cl < 3; cl++) {
        if (cls_len[cl] > 128 && c->p.max_readlen <= cls_len[cl - 1]) break;  // -L rules the longer classes out
        basal_align_extra ex;
        ex.order = d.order + (size_t)cl * mr;
        ex.n_ptr = &d.cnt->cls_n[cl];
        ex.ghost_base = mr;
        ex.counter = d.counter;
        ex.scratch = d.scratch;
        int rc = basal_launch_align(c, d.text, d.text_cap + 1024, d.desc, mr, d.stales, mr, cls_len[cl], smode, d.results, d.stream, d.stream_cap, &d.cnt->stream_used, s.comp, &ex);
        if (rc) return rc;
    }
    return BASAL_OK;
}

// what follows the align launches of a SAM batch: (paired-end: the pairing kernel over the mates' logs, then) the text
static int queue_format(basal_pipe *p, Slot &s) {
    PipeDev &v = p->devs[(size_t)s.dev];
    basal_core *c = v.c;
    SlotDev &d = s.d;
    if (!(p->o.flags & BASAL_PIPE_PAIRS)) return prep_enqueue_format(c, p->k, d, v.sh, p->o.max_reads, s.comp);
    const uint32_t npairs = s.n_host / 2;
    if (basal_fill_async(&d.cnt->pe[0], sizeof d.cnt->pe, 0, s.comp) || basal_fill_async(&d.cnt->pe_recs_used, sizeof d.cnt->pe_recs_used, 0, s.comp)) return BASAL_EDEVICE;
    int rc = basal_pe_enqueue(c, d.desc, d.results, d.stream, d.pe_work, npairs, d.pe_pairs, d.pe_recs, d.pe_recs_cap, &d.cnt->pe_recs_used, d.cnt->pe, s.comp);
    if (rc) return rc;
    return prep_enqueue_format_pe(c, p->k, d, v.sh, npairs, p->o.max_reads, s.comp);
}

static int submit_common(basal_pipe *p, int mode, uint64_t nbytes, uint32_t n, int format, uint32_t first_index, uint32_t readset, uint32_t max_len,
                         uint32_t pair_split = 0) {
    if (!p) { set_error("pipe_submit: null argument"); return BASAL_EINVAL; }
    int si;
    bool broken;
    {
        std::lock_guard<std::mutex> lk(p->m);
        si = p->acquired;
        broken = p->broken;
    }
    if (si < 0) { set_error("pipe_submit: no acquired slot (call basal_pipe_acquire first)"); return BASAL_ESTATE; }
    Slot &s = p->slots[(size_t)si];
    PipeDev &v = p->devs[(size_t)s.dev];
    basal_core *c = v.c;
    auto fail = [&](int rc) {
        std::lock_guard<std::mutex> lk(p->m);
        s.state = ST_FREE;
        p->acquired = -1;
        p->cv.notify_all();
        return rc;
    };
    if (broken) { set_error("pipe_submit: the pipe is stopped (a batch was refused): call basal_pipe_rewind"); return fail(BASAL_ESTATE); }
    // (the pipe may have been created while the reference was still being staged: page-locking its buffers takes a while)
    if (!c->have_ref || !c->have_index) { set_error("pipe_submit: stage the reference and the index first"); return fail(BASAL_ESTATE); }
    if (p->o.output == BASAL_PIPE_OUT_SAM && (!c->d_names || c->n_names != c->ncontig)) { set_error("pipe_submit: SAM output needs basal_core_set_contig_names first"); return fail(BASAL_ESTATE); }
    v.sh.names = c->d_names;
    v.sh.name_off = c->d_name_off;
    if (nbytes > p->o.max_bytes || n > p->o.max_reads) { set_error("pipe_submit: batch larger than the pipe's max_bytes / max_reads"); return fail(BASAL_EINVAL); }
    if ((p->o.flags & BASAL_PIPE_PAIRS) && (!(mode == MODE_RECORDS || (mode == MODE_TEXT && pair_split)) || (n & 1u) || n == 0)) {
        set_error("pipe_submit: a paired-end pipe takes submit_records with an even number of records (a0, b0, a1, b1, ...) or submit_text_pairs");
        return fail(BASAL_EINVAL);
    }
    if (pair_split && !(p->o.flags & BASAL_PIPE_PAIRS)) { set_error("pipe_submit_text_pairs: not a paired-end pipe"); return fail(BASAL_EINVAL); }
    if ((mode == MODE_PREPARED) != (p->o.output == BASAL_PIPE_OUT_RESULTS)) { set_error("pipe_submit: prepared reads go with BASAL_PIPE_OUT_RESULTS, text and records with BASAL_PIPE_OUT_SAM"); return fail(BASAL_EINVAL); }
    if (hipSetDevice(c->device) != hipSuccess) { set_error("pipe_submit: hipSetDevice failed"); return fail(BASAL_EDEVICE); }
    SlotDev &d = s.d;
    const uint32_t mr = p->o.max_reads;
    s.mode = mode; s.nbytes = nbytes; s.n_host = n; s.max_len = max_len; s.out_queued = false; s.out_bytes = 0;
    uint32_t bno;
    int src_dev, src_slot;
    {
        std::lock_guard<std::mutex> lk(p->m);
        bno = p->next_batch;
        src_dev = p->carry_dev[bno % p->ncarry];
        src_slot = p->carry_slot[bno % p->ncarry];
    }
    s.batch_no = bno;
    s.comp = (bno & 1u) && !getenv("BASAL_PIPE_ONE_STREAM") ? v.st_comp2 : v.st_comp;
#define TRYS(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return fail(BASAL_EDEVICE); } } while (0)
#define TRYR(x) do { int r_ = (x); if (r_) return fail(r_); } while (0)
    // copies in
    TRYS(hipEventRecord(s.ev[EV_START], v.st_in));
    if (nbytes) TRYS(hipMemcpyAsync(d.text, s.h_blob, nbytes, hipMemcpyHostToDevice, v.st_in));
    if (mode == MODE_RECORDS && n) TRYS(hipMemcpyAsync(d.raw, s.h_raw, (size_t)n * sizeof(basal_rawread), hipMemcpyHostToDevice, v.st_in));
    if (mode == MODE_PREPARED && n) TRYS(hipMemcpyAsync(d.raw, s.h_raw, (size_t)n * sizeof(basal_read), hipMemcpyHostToDevice, v.st_in));
    TRYS(hipEventRecord(s.ev[EV_H2D], v.st_in));
    // kernels (this stream also keeps the batches' carry states in order)
    TRYS(hipStreamWaitEvent(s.comp, s.ev[EV_H2D], 0));
    TRYS(hipEventRecord(s.ev[EV_COMP0], s.comp));
    if (basal_fill_async(d.cnt, sizeof(BatchCounters), 0, s.comp)) return BASAL_EDEVICE;
    // the state this batch starts from was written by the prep kernels of the batch before, on the other kernel stream or on another GPU: behind them
    if (mode != MODE_PREPARED && src_slot >= 0 && src_slot != si) TRYS(hipStreamWaitEvent(s.comp, p->slots[(size_t)src_slot].ev[EV_PREP], 0));
    if (mode != MODE_PREPARED && src_dev >= 0 && src_dev != s.dev) {  // ... and, from another GPU, fetched
        PipeDev &w = p->devs[(size_t)src_dev];
        TRYS(hipMemcpyPeerAsync(v.sh.carry[bno % p->ncarry], c->device, w.sh.carry[bno % p->ncarry], w.c->device, sizeof(CarryState), s.comp));
    }
    if (mode != MODE_PREPARED) {
        if (mode == MODE_TEXT) TRYR(prep_enqueue_index_text(c, d, v.sh, bno, nbytes, format, first_index, pair_split ? 0xFFFFFFFFu : p->read_end, readset, mr, pair_split, pair_split ? n / 2 : 0, s.comp));
        else TRYS(hipMemcpyAsync(&d.cnt->n_reads, &s.n_host, sizeof(uint32_t), hipMemcpyHostToDevice, s.comp));
        TRYR(prep_enqueue_filter(c, p->k, d, v.sh, bno, mr, true, 0, s.comp));
        if (p->o.flags & BASAL_PIPE_PAIRS) TRYR(prep_enqueue_pair_fix(c, d, n / 2, s.comp));
    }
    TRYS(hipEventRecord(s.ev[EV_PREP], s.comp));
    TRYR(queue_align(p, s));
    TRYS(hipEventRecord(s.ev[EV_ALIGN], s.comp));
    if (mode != MODE_PREPARED) TRYR(queue_format(p, s));
    TRYS(hipEventRecord(s.ev[EV_FORMAT], s.comp));
    // copies out: the counters (and the guard ledger) always; the output itself once its size is known
    TRYS(hipStreamWaitEvent(v.st_cnt, s.ev[EV_FORMAT], 0));
    TRYS(hipMemcpyAsync(s.h_cnt, d.cnt, sizeof(BatchCounters), hipMemcpyDeviceToHost, v.st_cnt));
    TRYS(hipMemcpyAsync(s.h_guard, d.counter + 1, 24 * sizeof(unsigned int), hipMemcpyDeviceToHost, v.st_cnt));
    if (basal_fill_async(d.counter + 1, 24 * sizeof(unsigned int), 0, v.st_cnt)) return BASAL_EDEVICE;
    TRYS(hipEventRecord(s.ev[EV_COUNTERS], v.st_cnt));
    if (mode == MODE_PREPARED) {  // the size of the output is known: queue its copy right behind the kernel
        TRYS(hipStreamWaitEvent(v.st_out, s.ev[EV_FORMAT], 0));
        TRYS(hipEventRecord(s.ev[EV_OUT0], v.st_out));
        if (n) TRYS(hipMemcpyAsync(s.h_out, d.results, (size_t)n * sizeof(basal_result), hipMemcpyDeviceToHost, v.st_out));
        s.out_bytes = (uint64_t)n * sizeof(basal_result);
        TRYS(hipEventRecord(s.ev[EV_OUT], v.st_out));
        s.out_queued = true;
    }
#undef TRYS
#undef TRYR
    {
        std::lock_guard<std::mutex> lk(p->m);
        s.state = ST_INFLIGHT;
        p->inflight.push_back(si);
        p->acquired = -1;
        p->next_batch = bno + 1;
        if (mode != MODE_PREPARED) {  // the state the next batch starts from: written here, by this slot's prep kernels
            p->carry_dev[(bno + 1) % p->ncarry] = s.dev;
            p->carry_slot[(bno + 1) % p->ncarry] = si;
        }
        // the batches in front may have finished meanwhile: start the copy of their output now rather than when they are collected
        for (int j : p->inflight)
            if (j != si && !p->slots[(size_t)j].collecting) try_queue_output(p, p->slots[(size_t)j], false);
    }
    return BASAL_OK;
}

extern "C" int basal_pipe_submit_text(basal_pipe_t *p, uint64_t nbytes, int format, uint32_t first_index, uint32_t readset) {
    if (format != BASAL_FMT_FASTQ && format != BASAL_FMT_FASTA) { set_error("pipe_submit_text: bad format"); return BASAL_EINVAL; }
    return submit_common(p, MODE_TEXT, nbytes, 0, format, first_index, readset, 0);
}
extern "C" int basal_pipe_submit_records(basal_pipe_t *p, uint64_t nblob, uint32_t n) { return submit_common(p, MODE_RECORDS, nblob, n, 0, 0, 0, 0); }
extern "C" int basal_pipe_submit_text_pairs(basal_pipe_t *p, uint64_t nbytes, uint64_t split, uint32_t npairs, int format, uint32_t first_index) {
    if (split == 0 || split >= nbytes || npairs == 0 || npairs > 0x7FFFFFFFu / 2) { set_error("pipe_submit_text_pairs: bad split / pair count"); if (p) basal_pipe_cancel(p); return BASAL_EINVAL; }
    return submit_common(p, MODE_TEXT, nbytes, 2 * npairs, format, first_index, 0, 0, (uint32_t)split);
}
extern "C" int basal_pipe_submit_prepared(basal_pipe_t *p, uint64_t nbases, uint32_t n, uint32_t max_len) {
    if (max_len == 0 || max_len > BASAL_MAXREADLEN) { set_error("pipe_submit_prepared: max_len must be 1..480"); return BASAL_EINVAL; }
    return submit_common(p, MODE_PREPARED, nbases, n, 0, 0, 0, max_len);
}

extern "C" int basal_pipe_set_read_range(basal_pipe_t *p, uint32_t next_index, uint32_t read_end) {
    if (!p) { set_error("pipe_set_read_range: null argument"); return BASAL_EINVAL; }
    std::lock_guard<std::mutex> lk(p->m);
    if (!p->inflight.empty() || p->acquired >= 0) { set_error("pipe_set_read_range: batches in flight"); return BASAL_ESTATE; }
    const uint32_t at = p->next_batch % p->ncarry;
    const int home = p->carry_dev[at];
    for (size_t g = 0; g < p->devs.size(); g++) {  // (the state the next batch starts from: on the GPU that holds it, or on all of them before the first batch)
        if (home >= 0 && (int)g != home) continue;
        PipeDev &v = p->devs[g];
        HIP_TRYQ(hipSetDevice(v.c->device));
        HIP_TRYQ(hipDeviceSynchronize());
        HIP_TRYQ(hipMemcpy(&v.sh.carry[at]->next_index, &next_index, sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    p->read_end = read_end;
    return BASAL_OK;
}

extern "C" int basal_pipe_cancel(basal_pipe_t *p) {
    if (!p) { set_error("pipe_cancel: null argument"); return BASAL_EINVAL; }
    std::lock_guard<std::mutex> lk(p->m);
    if (p->acquired >= 0) {
        p->slots[(size_t)p->acquired].state = ST_FREE;
        p->acquired = -1;
        p->cv.notify_all();
    }
    return BASAL_OK;
}

extern "C" int basal_pipe_stop(basal_pipe_t *p) {
    if (!p) { set_error("pipe_stop: null argument"); return BASAL_EINVAL; }
    std::lock_guard<std::mutex> lk(p->m);
    if (!p->broken) {
        p->broken = true;
        p->rewind_to = p->inflight.empty() ? p->next_batch : p->slots[(size_t)p->inflight.front()].batch_no;
    }
    p->cv.notify_all();
    return BASAL_OK;
}

extern "C" int basal_pipe_rewind(basal_pipe_t *p) {
    if (!p) { set_error("pipe_rewind: null argument"); return BASAL_EINVAL; }
    std::lock_guard<std::mutex> lk(p->m);
    for (auto &v : p->devs) {
        HIP_TRYQ(hipSetDevice(v.c->device));
        HIP_TRYQ(hipStreamSynchronize(v.st_in));
        HIP_TRYQ(hipStreamSynchronize(v.st_comp));
        HIP_TRYQ(hipStreamSynchronize(v.st_comp2));
        HIP_TRYQ(hipStreamSynchronize(v.st_out));
        HIP_TRYQ(hipStreamSynchronize(v.st_cnt));
    }
    for (auto &s : p->slots) {
        s.state = ST_FREE;
        s.collecting = false;
    }
    p->inflight.clear();
    p->acquired = p->held = -1;
    if (p->broken) p->next_batch = p->rewind_to;
    // (the state batch rewind_to started from is where it was: on the GPU that prepared the batch before it; everything is idle now, so no
    // event is pending for any state; the states of the dropped batches are simply written again)
    for (auto &cs : p->carry_slot) cs = -1;
    p->broken = false;
    p->cv.notify_all();
    return BASAL_OK;
}

extern "C" int basal_pipe_release(basal_pipe_t *p) {
    if (!p) { set_error("pipe_release: null argument"); return BASAL_EINVAL; }
    std::lock_guard<std::mutex> lk(p->m);
    if (p->held >= 0) {
        p->slots[(size_t)p->held].state = ST_FREE;
        p->held = -1;
        p->cv.notify_all();
    }
    return BASAL_OK;
}

extern "C" int basal_pipe_collect(basal_pipe_t *p, const void **out, uint64_t *nbytes, basal_batch_stats *stats) {
    if (!p) { set_error("pipe_collect: null argument"); return BASAL_EINVAL; }
    int si;
    {
        std::lock_guard<std::mutex> lk(p->m);
        if (p->held >= 0) {  // the output handed out last is no longer needed: its slot is free again
            p->slots[(size_t)p->held].state = ST_FREE;
            p->held = -1;
            p->cv.notify_all();
        }
        if (p->broken) { set_error("pipe_collect: the pipe is stopped (a batch was refused): call basal_pipe_rewind"); return BASAL_ESTATE; }
        if (p->inflight.empty()) { set_error("pipe_collect: nothing in flight"); return BASAL_ESTATE; }
        si = p->inflight.front();
        p->slots[(size_t)si].collecting = true;
    }
    Slot &s = p->slots[(size_t)si];
    PipeDev &v = p->devs[(size_t)s.dev];
    basal_core *c = v.c;
    HIP_TRYQ(hipSetDevice(c->device));
    HIP_TRYQ(hipEventSynchronize(s.ev[EV_COUNTERS]));
    int ret = basal_report_guard(s.h_guard);
    const bool sam = p->o.output == BASAL_PIPE_OUT_SAM;
    if (!ret && sam && s.mode == MODE_TEXT && (s.h_cnt->irregular & 1u)) {
        set_error("pipe: the text of this batch is not regular FASTQ/FASTA (blank lines, white space inside a line, a wrapped sequence, or more reads than max_reads): "
                  "parse it on the host and submit it with basal_pipe_submit_records");
        ret = BASAL_EIO;
    }
    if (!ret && sam && (s.h_cnt->irregular & 4u)) {
        set_error("pipe: batch " + std::to_string(s.batch_no) + " did not start from the state the batch before it left (carry hand-over out of order): internal error");
        ret = BASAL_EDEVICE;
    }
    if (!ret && sam && s.h_cnt->pair_err) {
        set_error("Error: Paired reads name not match (pair " + std::to_string(s.h_cnt->pair_err_at) + " of the batch): the reference exits here (pairs.cpp:501-504)");
        ret = BASAL_EINVAL;
    }
    if (!ret && sam) {
        // rare second passes, with the batch's inputs still on the device: a hit stream (-r 2, paired-end) or a text buffer that was too small
        SlotDev &d = s.d;
        for (int pass = 0; pass < 4; pass++) {
            const BatchCounters &cn = *s.h_cnt;
            const bool stream_small = d.stream && cn.stream_used > d.stream_cap, out_small = cn.out_bytes > d.out_cap;
            const bool recs_small = d.pe_recs && cn.pe_recs_used > d.pe_recs_cap;
            if (!stream_small && !out_small && !recs_small && !(cn.irregular & 2u)) break;
            HIP_TRYQ(hipStreamSynchronize(s.comp));
            HIP_TRYQ(hipStreamSynchronize(v.st_cnt));
            if (stream_small || (cn.irregular & 2u)) {
                hipFree(d.stream);
                d.stream = nullptr;
                d.stream_cap = cn.stream_used + cn.stream_used / 4 + 4096;
                HIP_TRYQ(hipMalloc(&d.stream, d.stream_cap * sizeof(basal_hit)));
                if (d.pe_work) {
                    hipFree(d.pe_work);
                    d.pe_work = nullptr;
                    d.pe_work_cap = d.stream_cap;
                    HIP_TRYQ(hipMalloc(&d.pe_work, d.pe_work_cap * sizeof(basal_hit)));
                }
            }
            if (recs_small) {
                hipFree(d.pe_recs);
                d.pe_recs = nullptr;
                d.pe_recs_cap = cn.pe_recs_used + cn.pe_recs_used / 4 + 4096;
                HIP_TRYQ(hipMalloc(&d.pe_recs, d.pe_recs_cap * sizeof(basal_pe_rec)));
            }
            if (out_small) {
                hipFree(d.out);
                d.out = nullptr;
                d.out_cap = cn.out_bytes + cn.out_bytes / 4 + 4096;
                HIP_TRYQ(hipMalloc(&d.out, d.out_cap));
            }
            // counters of the stages that run again
            BatchCounters z = cn;
            z.n_aligned = z.n_unique = z.n_multiple = 0; z.out_bytes = 0; z.irregular &= ~2u;
            if (stream_small || (cn.irregular & 2u)) z.stream_used = 0;
            HIP_TRYQ(hipMemcpyAsync(d.cnt, &z, sizeof z, hipMemcpyHostToDevice, s.comp));
            HIP_TRYQ(hipStreamSynchronize(s.comp));
            if (stream_small || (cn.irregular & 2u)) { int rc = queue_align(p, s); if (rc) return rc; }
            { int rc = queue_format(p, s); if (rc) return rc; }
            HIP_TRYQ(hipMemcpyAsync(s.h_cnt, d.cnt, sizeof(BatchCounters), hipMemcpyDeviceToHost, s.comp));
            HIP_TRYQ(hipMemcpyAsync(s.h_guard, d.counter + 1, 24 * sizeof(unsigned int), hipMemcpyDeviceToHost, s.comp));
            HIP_TRYQ(hipMemsetAsync(d.counter + 1, 0, 24 * sizeof(unsigned int), s.comp));
            HIP_TRYQ(hipStreamSynchronize(s.comp));
            if ((ret = basal_report_guard(s.h_guard))) break;
        }
    }
    if (!ret) {
        int rc = try_queue_output(p, s, true);
        if (rc) ret = rc;
        else if (!s.out_queued) { set_error("pipe_collect: output of the batch does not fit after regrowing"); ret = BASAL_EOVERFLOW; }
        else HIP_TRYQ(hipEventSynchronize(s.ev[EV_OUT]));
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        const BatchCounters &cn = *s.h_cnt;
        stats->n_reads = s.mode == MODE_PREPARED ? s.n_host : cn.n_reads;
        stats->n_aligned = cn.n_aligned; stats->n_unique = cn.n_unique; stats->n_multiple = cn.n_multiple; stats->n_filtered = cn.n_filtered;
        for (int k = 0; k < 9; k++) stats->pe[k] = cn.pe[k];
        if (!ret) {
            hipEventElapsedTime(&stats->ms_h2d, s.ev[EV_START], s.ev[EV_H2D]);
            hipEventElapsedTime(&stats->ms_prep, s.ev[EV_COMP0], s.ev[EV_PREP]);
            hipEventElapsedTime(&stats->ms_align, s.ev[EV_PREP], s.ev[EV_ALIGN]);
            hipEventElapsedTime(&stats->ms_format, s.ev[EV_ALIGN], s.ev[EV_FORMAT]);
            hipEventElapsedTime(&stats->ms_d2h, s.ev[EV_OUT0], s.ev[EV_OUT]);
        }
    }
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->inflight.pop_front();
        s.collecting = false;
        if (ret == BASAL_EIO) {
            // the batches submitted after this one were prepared as if it held no reads: the pipe stops here until the caller has
            // quietened its submitter and called basal_pipe_rewind (the carry state this batch started from is still in the ring)
            p->broken = true;
            p->rewind_to = s.batch_no;
        }
        if (ret) s.state = ST_FREE;
        else { s.state = ST_HELD; p->held = si; }
        // start the output copy of the next finished batch while the caller works on this one
        if (!p->inflight.empty()) try_queue_output(p, p->slots[(size_t)p->inflight.front()], false);
        p->cv.notify_all();
    }
    if (ret) return ret;
    if (out) *out = s.h_out;
    if (nbytes) *nbytes = s.out_bytes;
    return BASAL_OK;
}
