// khg_host_comm.inl -- the tiled mode with its collectives inside the library (included by khugle_hip.hip).
//
// khg_tile_encode / khg_tile_decode run the per-GPU steps of khg_host_tile.inl and, between them, the exchanges the
// steps need, on a communicator (khg_comm):
//   * RCCL: one process per GPU, collectives over xGMI.  rccl.h supplies the types; the functions are looked up in
//     the RCCL library the process already holds (a PyTorch process brings its own copy), else in librccl.so.1, when
//     the first communicator is made, so the library loads and works on a machine without RCCL.
//   * local: the ranks are threads of one process and their contexts live on one GPU; the "collectives" are
//     device-to-device copies and one reduction kernel behind a host barrier.  For tests and rehearsals: it runs
//     the same host code as RCCL with eight ranks on a box that has one GPU.
// The reference has no counterpart: its one CompressImage call (Compress.h:104) transforms the whole picture.
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

struct KhgRcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    bool ok;
};

static KhgRcclApi* khg_rccl_api()
{
    static KhgRcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        memset(&api, 0, sizeof api);
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
        api.Reduce = (decltype(api.Reduce))dlsym(h, "ncclReduce");
        api.Send = (decltype(api.Send))dlsym(h, "ncclSend");
        api.Recv = (decltype(api.Recv))dlsym(h, "ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Reduce && api.Send &&
                 api.Recv && api.GroupStart && api.GroupEnd;
    });
    return &api;
}

// ---- the local group: n threads of one process, contexts on one device

enum { KHG_COMM_MAX = 8 };
enum KhgRedOp { KHG_RED_SUM_U32 = 0, KHG_RED_MIN_U32 = 1, KHG_RED_MIN_I32 = 2 };

struct KhgP2P {
    int peer;
    void* ptr;
    size_t bytes;
    bool send;
};

struct KhgLocalGroup {
    int n;
    std::mutex m;
    std::condition_variable cv;
    int arrived;
    uint64_t gen;
    int refs;
    bool broken;
    void* buf[KHG_COMM_MAX];                  // what each rank brought to the collective in progress
    std::vector<KhgP2P> posted[KHG_COMM_MAX];  // ... and to the point-to-point exchange in progress
};

// False when a rank has not arrived within two minutes (it failed outside the library, or never called): the group
// is broken from then on and every wait on it fails at once, so a test ends instead of hanging.
static bool khg_local_barrier(KhgLocalGroup* g)
{
    std::unique_lock<std::mutex> lk(g->m);
    if (g->broken) return false;
    const uint64_t gen = g->gen;
    if (++g->arrived == g->n) {
        g->arrived = 0;
        ++g->gen;
        g->cv.notify_all();
        return true;
    }
    if (!g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->gen != gen || g->broken; }) || g->broken) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return true;
}

struct KhgLocalPtrs {
    uint32_t* p[KHG_COMM_MAX];
};

// out[i] = the reduction of the n ranks' word i, written to rank `root`'s buffer (root < 0: to every rank's).
__global__ __launch_bounds__(256) void khg_local_reduce_kernel(KhgLocalPtrs b, int n, size_t count, int op, int root)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        uint32_t v = b.p[0][i];
        for (int r = 1; r < n; ++r) {
            const uint32_t w = b.p[r][i];
            if (op == KHG_RED_SUM_U32) v += w;
            else if (op == KHG_RED_MIN_U32) v = w < v ? w : v;
            else v = (int32_t)w < (int32_t)v ? w : v;
        }
        if (root >= 0) b.p[root][i] = v;
        else
            for (int r = 0; r < n; ++r) b.p[r][i] = v;
    }
}

struct khg_comm {
    int rank, n;
    ncclComm_t nccl;      // RCCL communicators
    bool own_nccl;        // ... made by khg_comm_create_rccl (destroyed with the communicator)
    KhgLocalGroup* local; // local groups
    // device memory of the gather merge, grown on demand: the packing plans, and the packed words sent or received
    void *plan, *data;
    size_t plan_bytes, data_bytes;
};

#define KHG_NCCL(call)                                   \
    do {                                                 \
        if ((call) != ncclSuccess) return KHG_ERR_HIP;   \
    } while (0)

// In-place reduction of `count` 32-bit words over the ranks, enqueued on (RCCL) or finished on (local) `st`.
// root < 0: to every rank.
static int khg_comm_reduce(khg_comm* cm, void* d_buf, size_t count, int op, int root, hipStream_t st)
{
    if (!count) return KHG_OK;
    if (cm->nccl) {
        KhgRcclApi* R = khg_rccl_api();
        const ncclDataType_t ty = op == KHG_RED_MIN_I32 ? ncclInt32 : ncclUint32;
        const ncclRedOp_t ro = op == KHG_RED_SUM_U32 ? ncclSum : ncclMin;
        if (root < 0) KHG_NCCL(R->AllReduce(d_buf, d_buf, count, ty, ro, cm->nccl, st));
        else KHG_NCCL(R->Reduce(d_buf, d_buf, count, ty, ro, root, cm->nccl, st));
        return KHG_OK;
    }
    KhgLocalGroup* g = cm->local;
    KHG_HIP(hipStreamSynchronize(st));  // this rank's buffer is complete
    g->buf[cm->rank] = d_buf;
    if (!khg_local_barrier(g)) return KHG_ERR_HIP;
    int rc = KHG_OK;
    if (cm->rank == 0) {
        KhgLocalPtrs b;
        for (int r = 0; r < g->n; ++r) b.p[r] = (uint32_t*)g->buf[r];
        const size_t blocks = (count + 255) / 256;
        hipLaunchKernelGGL(khg_local_reduce_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, b, g->n,
                           count, op, root);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = KHG_ERR_HIP;
    }
    if (!khg_local_barrier(g)) return KHG_ERR_HIP;  // the result is in place
    return rc;
}

// A batch of sends and receives; between two ranks they are matched in list order.
static int khg_comm_exchange(khg_comm* cm, const std::vector<KhgP2P>& ops, hipStream_t st)
{
    if (cm->nccl) {
        if (ops.empty()) return KHG_OK;
        KhgRcclApi* R = khg_rccl_api();
        KHG_NCCL(R->GroupStart());
        for (const KhgP2P& o : ops) {
            if (o.send) KHG_NCCL(R->Send(o.ptr, o.bytes, ncclUint8, o.peer, cm->nccl, st));
            else KHG_NCCL(R->Recv(o.ptr, o.bytes, ncclUint8, o.peer, cm->nccl, st));
        }
        KHG_NCCL(R->GroupEnd());
        return KHG_OK;
    }
    KhgLocalGroup* g = cm->local;
    KHG_HIP(hipStreamSynchronize(st));  // what this rank sends is complete
    g->posted[cm->rank] = ops;
    if (!khg_local_barrier(g)) return KHG_ERR_HIP;
    int rc = KHG_OK;
    size_t taken[KHG_COMM_MAX] = {0};  // sends of each peer to this rank already matched
    for (const KhgP2P& o : ops) {
        if (o.send) continue;
        const std::vector<KhgP2P>& theirs = g->posted[o.peer];
        size_t& i = taken[o.peer];
        while (i < theirs.size() && !(theirs[i].send && theirs[i].peer == cm->rank)) ++i;
        if (i == theirs.size() || theirs[i].bytes != o.bytes) {
            rc = KHG_ERR_ARG;  // the two ranks disagree about the exchange
            break;
        }
        if (hipMemcpyAsync(o.ptr, theirs[i].ptr, o.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = KHG_ERR_HIP;
        ++i;
    }
    if (hipStreamSynchronize(st) != hipSuccess) rc = KHG_ERR_HIP;
    if (!khg_local_barrier(g)) return KHG_ERR_HIP;  // every rank has taken what it was sent
    return rc;
}

extern "C" int khg_comm_unique_id(void* id, size_t bytes)
{
    if (!id || bytes < sizeof(ncclUniqueId)) return KHG_ERR_ARG;
    KhgRcclApi* R = khg_rccl_api();
    if (!R->ok) return KHG_ERR_HIP;
    ncclUniqueId u;
    KHG_NCCL(R->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return KHG_OK;
}

extern "C" int khg_comm_create_rccl(khg_comm** out, int rank, int nranks, const void* id, size_t bytes, int device)
{
    if (!out || !id || bytes < sizeof(ncclUniqueId) || nranks < 1 || nranks > KHG_COMM_MAX || rank < 0 || rank >= nranks)
        return KHG_ERR_ARG;
    KhgRcclApi* R = khg_rccl_api();
    if (!R->ok) return KHG_ERR_HIP;
    KHG_HIP(hipSetDevice(device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c;
    KHG_NCCL(R->CommInitRank(&c, nranks, u, rank));
    khg_comm* cm = new (std::nothrow) khg_comm{rank, nranks, c, true, nullptr, nullptr, nullptr, 0, 0};
    if (!cm) {
        (void)R->CommDestroy(c);
        return KHG_ERR_NOMEM;
    }
    *out = cm;
    return KHG_OK;
}

extern "C" int khg_comm_wrap_rccl(khg_comm** out, void* nccl_comm, int rank, int nranks)
{
    if (!out || !nccl_comm || nranks < 1 || nranks > KHG_COMM_MAX || rank < 0 || rank >= nranks) return KHG_ERR_ARG;
    if (!khg_rccl_api()->ok) return KHG_ERR_HIP;
    khg_comm* cm = new (std::nothrow) khg_comm{rank, nranks, (ncclComm_t)nccl_comm, false, nullptr, nullptr, nullptr, 0, 0};
    if (!cm) return KHG_ERR_NOMEM;
    *out = cm;
    return KHG_OK;
}

// A local group: khg_comm_local_group makes it, every rank's thread joins it with khg_comm_create_local.
extern "C" void* khg_comm_local_group(int nranks)
{
    if (nranks < 1 || nranks > KHG_COMM_MAX) return nullptr;
    KhgLocalGroup* g = new (std::nothrow) KhgLocalGroup;
    if (!g) return nullptr;
    g->n = nranks;
    g->arrived = 0;
    g->gen = 0;
    g->refs = nranks;
    g->broken = false;
    return g;
}

extern "C" int khg_comm_create_local(khg_comm** out, void* group, int rank)
{
    KhgLocalGroup* g = (KhgLocalGroup*)group;
    if (!out || !g || rank < 0 || rank >= g->n) return KHG_ERR_ARG;
    khg_comm* cm = new (std::nothrow) khg_comm{rank, g->n, nullptr, false, g, nullptr, nullptr, 0, 0};
    if (!cm) return KHG_ERR_NOMEM;
    *out = cm;
    return KHG_OK;
}

extern "C" void khg_comm_destroy(khg_comm* cm)
{
    if (!cm) return;
    if (cm->plan) (void)hipFree(cm->plan);
    if (cm->data) (void)hipFree(cm->data);
    if (cm->nccl && cm->own_nccl) (void)khg_rccl_api()->CommDestroy(cm->nccl);
    if (cm->local) {
        bool last;
        {
            std::lock_guard<std::mutex> lk(cm->local->m);
            last = --cm->local->refs == 0;
        }
        if (last) delete cm->local;
    }
    delete cm;
}

// The strip a rank transforms: its own rows with the neighbours' 24 edge rows attached (khg_tile_input_rows).
// d_own_*: the rank's rows / rows * nranks image rows; d_strip_*: room for the strip.
extern "C" int khg_tile_exchange_halos(khg_comm* cm, const uint8_t* d_own_r, const uint8_t* d_own_g, const uint8_t* d_own_b,
                                       int rows, int cols, uint8_t* d_strip_r, uint8_t* d_strip_g, uint8_t* d_strip_b,
                                       void* stream)
{
    if (!cm || !d_own_r || !d_own_g || !d_own_b || !d_strip_r || !d_strip_g || !d_strip_b) return KHG_ERR_ARG;
    const KhgTile t = {cm->rank, cm->n};
    if (!khg_dims_ok(rows, cols) || !khg_tile_ok(rows, t)) return KHG_ERR_DIMS;
    const int own = rows / cm->n;
    if (own < KHG_TILE_HALO) return KHG_ERR_DIMS;
    hipStream_t st = (hipStream_t)stream;
    const size_t halo = (size_t)KHG_TILE_HALO * cols, body = (size_t)own * cols;
    const size_t top = cm->rank > 0 ? halo : 0;  // the strip begins with the upper neighbour's rows
    const uint8_t* src[3] = {d_own_r, d_own_g, d_own_b};
    uint8_t* dst[3] = {d_strip_r, d_strip_g, d_strip_b};
    std::vector<KhgP2P> ops;
    for (int p = 0; p < 3; ++p) {
        KHG_HIP(hipMemcpyAsync(dst[p] + top, src[p], body, hipMemcpyDeviceToDevice, st));
        if (cm->rank > 0) {
            ops.push_back(KhgP2P{cm->rank - 1, (void*)src[p], halo, true});
            ops.push_back(KhgP2P{cm->rank - 1, dst[p], halo, false});
        }
        if (cm->rank < cm->n - 1) {
            ops.push_back(KhgP2P{cm->rank + 1, (void*)(src[p] + body - halo), halo, true});
            ops.push_back(KhgP2P{cm->rank + 1, dst[p] + top + body, halo, false});
        }
    }
    return khg_comm_exchange(cm, ops, st);
}

// ---- the merge that moves every byte of the file once.
//
// After the pack step every rank knows every chunk's bit count and bit offset, and which rank owns a chunk follows
// from the geometry (khg_chunk_info), so sender and keeper can both work out, without talking, which 32-bit words of
// the file a rank's chunks touch and where each lands in a packed send buffer: the words of the rank's chunks in
// file order, each word once (consecutive chunks of one rank share the word they meet in), for rank 0 with the
// header's 9 words in front and everything from the streams' last word on (pad bits, tables) behind.  Where chunks
// of two ranks meet inside a word both send it; their bits are disjoint and the keeper adds.

struct KhgGatherPlan {   // one rank's packing: entries 0 .. nchunks-1 the chunks, nchunks the header, nchunks+1 the tail
    uint32_t* start;     // first word in the file
    uint32_t* pos;       // ... and in the packed buffer
    uint32_t* cnt;       // words (0: nothing of this rank's)
    uint32_t* total;     // [1] words in the packed buffer
};

struct KhgGatherPlans {
    KhgGatherPlan of[KHG_COMM_MAX];
    int rank[KHG_COMM_MAX];
};

struct KhgGatherSmem {
    uint32_t scan[1024];
};

// Inclusive scan over the workgroup's 1024 threads; MAX or SUM.
template <bool MAX>
__device__ uint32_t khg_gather_scan(uint32_t v, KhgGatherSmem& sh)
{
    const uint32_t tid = threadIdx.x;
    sh.scan[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t o = tid >= d ? sh.scan[tid - d] : 0u;  // the neutral element of both
        __syncthreads();
        v = MAX ? (o > v ? o : v) : v + o;
        sh.scan[tid] = v;
        __syncthreads();
    }
    return v;
}

// Workgroup b plans rank pl.rank[b].  Chunks lie in the file in index order, so the end words are non-decreasing
// and "the words before this chunk that the rank has already taken" end at the end word of its previous chunk.
__global__ __launch_bounds__(1024) void khg_gather_plan_kernel(KhgEncArgs a, KhgGatherPlans pl)
{
    __shared__ KhgGatherSmem sh;
    const KhgGatherPlan g = pl.of[blockIdx.x];
    a.rank = pl.rank[blockIdx.x];
    const uint32_t tid = threadIdx.x, n = a.nchunks;
    const uint32_t seg = (n + 1023) / 1024, c0 = tid * seg < n ? tid * seg : n, c1 = c0 + seg < n ? c0 + seg : n;
    const uint32_t head = a.rank == 0 ? 9u : 0u;  // Writer.h:9-23: the 36-byte header
    auto w0 = [&](uint32_t c) { return (uint32_t)(a.chunk_off[c] >> 5); };
    auto w1 = [&](uint32_t c) { return (uint32_t)((a.chunk_off[c] + a.chunk_bits[c] + 31) >> 5); };
    auto takes = [&](uint32_t c) { return a.chunk_bits[c] != 0 && khg_chunk_info(a, c).mine; };
    uint32_t last = 0;
    for (uint32_t c = c0; c < c1; ++c)
        if (takes(c)) last = w1(c);
    const uint32_t incl = khg_gather_scan<true>(last, sh);
    __syncthreads();
    uint32_t before = tid ? sh.scan[tid - 1] : 0u;  // end word of the rank's last chunk before this segment
    if (before < head) before = head;
    __syncthreads();
    uint32_t sum = 0, e = before;
    for (uint32_t c = c0; c < c1; ++c)
        if (takes(c)) {
            const uint32_t s = w0(c) > e ? w0(c) : e;
            e = w1(c);
            sum += e - s;
        }
    const uint32_t upto = khg_gather_scan<false>(sum, sh);
    uint32_t at = head + upto - sum;
    e = before;
    for (uint32_t c = c0; c < c1; ++c) {
        uint32_t s = 0, k = 0;
        if (takes(c)) {
            s = w0(c) > e ? w0(c) : e;
            e = w1(c);
            k = e - s;
        }
        g.start[c] = s;
        g.pos[c] = at;
        g.cnt[c] = k;
        at += k;
    }
    if (tid == 1023) {
        const uint32_t nwords = (uint32_t)((a.res->total_len + 3) / 4);
        const uint32_t all_end = incl > head ? incl : head;  // the rank's words so far end here
        uint32_t ts = n ? (uint32_t)((a.chunk_off[n - 1] + a.chunk_bits[n - 1]) >> 5) : 9u;
        if (ts < all_end) ts = all_end;
        const uint32_t tail = a.rank == 0 && nwords > ts ? nwords - ts : 0u;
        g.start[n] = 0;
        g.pos[n] = 0;
        g.cnt[n] = head;
        g.start[n + 1] = ts;
        g.pos[n + 1] = head + upto;
        g.cnt[n + 1] = tail;
        *g.total = head + upto + tail;
    }
}

// ADD: file[start + i] += packed[pos + i] (the keeper); else packed[pos + i] = file[start + i] (a sender).
// A wave takes 64 entries at a time and its lanes copy the words of those that have any.
template <bool ADD>
__global__ __launch_bounds__(256) void khg_gather_move_kernel(KhgGatherPlan g, uint32_t entries, uint32_t* file, uint32_t* packed)
{
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwave = gridDim.x * 4;
    for (uint32_t base = wave * 64; base < entries; base += nwave * 64) {
        const uint32_t i = base + lane;
        const uint32_t k = i < entries ? g.cnt[i] : 0u, s = i < entries ? g.start[i] : 0u, p = i < entries ? g.pos[i] : 0u;
        uint64_t some = __ballot(k != 0);
        while (some) {
            const int l = __ffsll((long long)some) - 1;
            some &= some - 1;
            const uint32_t kk = __shfl(k, l), ss = __shfl(s, l), pp = __shfl(p, l);
            for (uint32_t w = lane; w < kk; w += 64) {
                if (ADD) file[ss + w] += packed[pp + w];
                else packed[pp + w] = file[ss + w];
            }
        }
    }
}

static int khg_comm_room(void** buf, size_t* have, size_t want)
{
    if (want <= *have) return KHG_OK;
    if (*buf) KHG_HIP(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    if (hipMalloc(buf, want) != hipSuccess) return KHG_ERR_NOMEM;
    *have = want;
    return KHG_OK;
}

// d_out holds this rank's share (khg_tile_pack_dev); on return the file on rank `root`.
static int khg_tile_gather(khg_ctx* c, khg_comm* cm, uint8_t* d_out, size_t cap, int root, hipStream_t st)
{
    const KhgEncArgs e = khg_enc_args(c->ws, c->tile_rows, c->tile_cols, c->tile_lvl, d_out, cap, c->tile);
    const size_t entries = (size_t)e.nchunks + 2;
    const bool keeper = cm->rank == root;
    const int nplan = keeper ? cm->n : 1;
    const size_t per = khg_align(entries * 4);
    int rc;
    if ((rc = khg_comm_room(&cm->plan, &cm->plan_bytes, (size_t)nplan * 3 * per + 256))) return rc;
    KhgGatherPlans pl;
    memset(&pl, 0, sizeof pl);
    char* base = (char*)cm->plan;
    uint32_t* totals = (uint32_t*)base;
    for (int i = 0; i < nplan; ++i) {
        pl.rank[i] = keeper ? i : cm->rank;
        pl.of[i].start = (uint32_t*)(base + 256 + (size_t)(3 * i) * per);
        pl.of[i].pos = (uint32_t*)(base + 256 + (size_t)(3 * i + 1) * per);
        pl.of[i].cnt = (uint32_t*)(base + 256 + (size_t)(3 * i + 2) * per);
        pl.of[i].total = totals + i;
    }
    hipLaunchKernelGGL(khg_gather_plan_kernel, dim3(nplan), dim3(1024), 0, st, e, pl);
    KHG_HIP(hipGetLastError());
    uint32_t total[KHG_COMM_MAX];
    KHG_HIP(hipMemcpyAsync(total, totals, (size_t)nplan * 4, hipMemcpyDeviceToHost, st));
    KHG_HIP(hipStreamSynchronize(st));
    size_t words = 0, at[KHG_COMM_MAX];
    for (int i = 0; i < nplan; ++i) {
        at[i] = words;
        if (!keeper || i != root) words += ((size_t)total[i] + 63) & ~(size_t)63;
    }
    if ((rc = khg_comm_room(&cm->data, &cm->data_bytes, words * 4 + 256))) return rc;
    uint32_t* data = (uint32_t*)cm->data;
    const unsigned grid = (unsigned)((entries + 255) / 256 < 2048 ? (entries + 255) / 256 : 2048);
    std::vector<KhgP2P> ops;
    if (!keeper) {
        hipLaunchKernelGGL(khg_gather_move_kernel<false>, dim3(grid), dim3(256), 0, st, pl.of[0], (uint32_t)entries,
                           (uint32_t*)d_out, data);
        KHG_HIP(hipGetLastError());
        ops.push_back(KhgP2P{root, data, (size_t)total[0] * 4, true});
        return khg_comm_exchange(cm, ops, st);
    }
    for (int r = 0; r < cm->n; ++r)
        if (r != root) ops.push_back(KhgP2P{r, data + at[r], (size_t)total[r] * 4, false});
    if ((rc = khg_comm_exchange(cm, ops, st))) return rc;
    for (int r = 0; r < cm->n; ++r)  // one rank after the other: within a rank's buffer every word of the file appears once
        if (r != root)
            hipLaunchKernelGGL(khg_gather_move_kernel<true>, dim3(grid), dim3(256), 0, st, pl.of[r], (uint32_t)entries,
                               (uint32_t*)d_out, data + at[r]);
    KHG_HIP(hipGetLastError());
    return KHG_OK;
}

// What the ranks' host-side checks said, agreed: the lowest (worst) code of all ranks, on all ranks.
static int khg_comm_agree(khg_ctx* c, khg_comm* cm, int mine, hipStream_t st, int* agreed)
{
    int32_t* d = &c->ws.res->pad_;
    int32_t v = mine;
    KHG_HIP(hipMemcpyAsync(d, &v, 4, hipMemcpyHostToDevice, st));
    int rc = khg_comm_reduce(cm, d, 1, KHG_RED_MIN_I32, -1, st);
    if (rc) return rc;
    KHG_HIP(hipMemcpyAsync(&v, d, 4, hipMemcpyDeviceToHost, st));
    KHG_HIP(hipStreamSynchronize(st));
    *agreed = v;
    return KHG_OK;
}

// All ranks call this with their strip.  The .comp bytes -- those of one encoder -- arrive in d_out on rank `root`
// (the other ranks' d_out holds their share: scratch); *out_len is the file's length on every rank.
// merge 0: the ranks' zero-padded shares are added up in one reduce of the file's size; 1: every byte of the file
// moves once, point to point (khg_tile_gather).
// A failure on any rank -- an argument its context refuses, a symbol out of range, a stream of 2^31 bits, an output
// that does not fit -- is returned by all of them: the ranks agree on what their checks said before the first
// collective on image data and on the kernels' status before the merge, so none waits in a collective that its
// peers never enter.  (A HIP or RCCL failure is returned at once: the communicator is of no use after it.)
// Waits for the stream.
extern "C" int khg_tile_encode(khg_ctx* c, khg_comm* cm, const uint8_t* d_r, const uint8_t* d_g, const uint8_t* d_b,
                               int rows, int cols, int lvl, uint8_t* d_out, size_t cap, size_t* out_len, int root,
                               int merge, void* stream)
{
    if (!c || !cm || !out_len || root < 0 || root >= cm->n || merge < 0 || merge > 1) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc, mine = KHG_OK, agreed = KHG_OK;
    if (!d_out || ((uintptr_t)d_out & 3)) mine = KHG_ERR_ARG;
    else if (cap < 40) mine = KHG_ERR_CAP;
    else mine = khg_tile_forward_dev(c, d_r, d_g, d_b, rows, cols, lvl, cm->rank, cm->n, stream);
    if (mine == KHG_ERR_HIP) return mine;
    if ((rc = khg_comm_agree(c, cm, mine, st, &agreed))) return rc;
    if (agreed != KHG_OK) return mine != KHG_OK ? mine : agreed;
    if ((rc = khg_comm_reduce(cm, c->ws.hist_cnt, 3 * KHG_NBINS, KHG_RED_SUM_U32, -1, st))) return rc;
    // "absent" is 0xFFFFFFFF, the largest value: the minimum over the ranks is the first raster index
    if ((rc = khg_comm_reduce(cm, c->ws.hist_first, 3 * KHG_NBINS, KHG_RED_MIN_U32, -1, st))) return rc;
    if ((rc = khg_tile_lens_dev(c, stream))) return rc;
    const size_t nchunks = khg_tile_chunk_count(rows, cols, cm->n);
    if ((rc = khg_comm_reduce(cm, c->ws.chunk_bits, nchunks, KHG_RED_SUM_U32, -1, st))) return rc;
    if ((rc = khg_tile_pack_dev(c, d_out, cap, stream))) return rc;
    if ((rc = khg_comm_reduce(cm, &c->ws.res->status, 1, KHG_RED_MIN_I32, -1, st))) return rc;
    KHG_HIP(hipMemcpyAsync(c->h_res, c->ws.res, sizeof(KhgResult), hipMemcpyDeviceToHost, st));
    KHG_HIP(hipStreamSynchronize(st));
    c->pending = 0;
    if (c->h_res->status != 0) return c->h_res->status;
    const size_t total = (size_t)c->h_res->total_len;
    if (total >= ((size_t)1 << 34)) return KHG_ERR_RANGE;  // word indices are 32-bit
    if (merge == 1) rc = khg_tile_gather(c, cm, d_out, cap, root, st);
    else rc = khg_comm_reduce(cm, d_out, (total + 3) / 4, KHG_RED_SUM_U32, root, st);  // disjoint bits: the sum is the OR
    if (rc) return rc;
    KHG_HIP(hipStreamSynchronize(st));
    *out_len = total;
    return KHG_OK;
}

// All ranks call this with the whole file.  d_r / d_g / d_b receive the rank's rows of the picture.  The entropy
// decode is split over the ranks (khg_tile_decode_map_dev ... khg_tile_decode_inverse_dev) and the decoded symbols
// go point to point from the ranks that hold them straight into the symbol planes of the ranks whose rows need
// them.  Waits for the stream.
extern "C" int khg_tile_decode(khg_ctx* c, khg_comm* cm, const uint8_t* d_comp, size_t len, int rows, int cols,
                               uint8_t* d_r, uint8_t* d_g, uint8_t* d_b, void* stream)
{
    if (!c || !cm) return KHG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int n = cm->n, me = cm->rank;
    int rc;
    int mine = khg_tile_decode_map_dev(c, d_comp, len, rows, cols, me, n, stream), agreed = KHG_OK;
    if (mine == KHG_ERR_HIP) return mine;
    if (mine == KHG_OK && (!d_r || !d_g || !d_b || ((uintptr_t)d_r & 7) || ((uintptr_t)d_g & 7) || ((uintptr_t)d_b & 7)))
        mine = KHG_ERR_ARG;
    if ((rc = khg_comm_agree(c, cm, mine, st, &agreed))) return rc;
    if (agreed != KHG_OK) {
        c->tdec_step = 0;
        return mine != KHG_OK ? mine : agreed;
    }
    const KhgDecArgs d = khg_tdec_args(c);
    for (int p = 0; p < 3; ++p)  // other ranks' records are zero here: the sum is the union
        if ((rc = khg_comm_reduce(cm, c->dws.blk + d.blk0[p], (size_t)c->tdec_nb[p] * (sizeof(KhgBlkInfo) / 4),
                                  KHG_RED_SUM_U32, -1, st)))
            return rc;
    std::vector<uint32_t> bounds(3 * (size_t)(n + 1));
    if ((rc = khg_tile_decode_write_dev(c, bounds.data(), stream))) return rc;  // identical on every rank
    std::vector<KhgP2P> ops;
    for (int dst = 0; dst < n; ++dst)
        for (int p = 0; p < 3; ++p) {
            uint32_t need[4][2];
            const int nr = khg_tile_need_ranges_impl(rows, cols, KhgTile{dst, n}, p, need);
            for (int i = 0; i < nr; ++i)
                for (int src = 0; src < n; ++src) {
                    if (src == dst) continue;
                    const uint32_t b0 = bounds[p * (n + 1) + src], b1 = bounds[p * (n + 1) + src + 1];
                    const uint32_t lo = need[i][0] > b0 ? need[i][0] : b0, hi = need[i][1] < b1 ? need[i][1] : b1;
                    if (lo >= hi) continue;
                    if (src == me) ops.push_back(KhgP2P{dst, c->ws.sym[p] + lo, (size_t)(hi - lo) * 2, true});
                    else if (dst == me) ops.push_back(KhgP2P{src, c->ws.sym[p] + lo, (size_t)(hi - lo) * 2, false});
                }
        }
    if ((rc = khg_comm_exchange(cm, ops, st))) return rc;
    if ((rc = khg_tile_decode_inverse_dev(c, d_r, d_g, d_b, stream))) return rc;
    int32_t* d_st = &c->ws.res->status;
    // a damaged stream may show on the rank that decoded that part only: agree before anyone reports
    if ((rc = khg_comm_reduce(cm, d_st, 1, KHG_RED_MIN_I32, -1, st))) return rc;
    KHG_HIP(hipMemcpyAsync(c->h_res, c->ws.res, sizeof(KhgResult), hipMemcpyDeviceToHost, st));
    KHG_HIP(hipStreamSynchronize(st));
    c->pending = 0;
    return c->h_res->status;
}
