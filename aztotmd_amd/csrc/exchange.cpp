#include "exchange.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

namespace aztot {

void check_hip(hipError_t e, const char* what);
#define HIP_CHECK(x) check_hip((x), #x)

// ---------------------------------------------------------------------------------------------------
// host-staged transport (tests over gloo): D2H, caller's sendrecv, H2D.  The header at the front of each
// message tells how many bytes are really in use, so only those travel.
// ---------------------------------------------------------------------------------------------------
void CallbackExchanger::exchange(int left, int right, const void* dSendLeft, const void* dSendRight, void* dFromLeft, void* dFromRight,
                                 size_t bytes, hipStream_t stream)
{
    for (int k = 0; k < 2; k++) { hs_[k].resize(bytes); hr_[k].resize(bytes); }
    HIP_CHECK(hipMemcpyAsync(hs_[0].data(), dSendLeft, bytes, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(hs_[1].data(), dSendRight, bytes, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    int64_t got = 0;
    // leftward messages first (everyone sends left / receives from the right), then rightward ones
    if (sr_(ctx_, left, hs_[0].data(), (int64_t)bytes, right, hr_[1].data(), (int64_t)bytes, &got) != 0)
        throw std::runtime_error("slab exchange callback failed (leftward)");
    if (sr_(ctx_, right, hs_[1].data(), (int64_t)bytes, left, hr_[0].data(), (int64_t)bytes, &got) != 0)
        throw std::runtime_error("slab exchange callback failed (rightward)");
    HIP_CHECK(hipMemcpyAsync(dFromLeft, hr_[0].data(), bytes, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(dFromRight, hr_[1].data(), bytes, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

void CallbackExchanger::exchange_ranges(int left, int right, double* const* arrays, int nArr, int sLb, int sLn, int sRb, int sRn, int rLb, int rLn,
                                        int rRb, int rRn, hipStream_t stream)
{
    const size_t bsL = sizeof(double) * (size_t)nArr * sLn, bsR = sizeof(double) * (size_t)nArr * sRn;
    const size_t brL = sizeof(double) * (size_t)nArr * rLn, brR = sizeof(double) * (size_t)nArr * rRn;
    hs_[0].resize(std::max<size_t>(bsL, 8)); hs_[1].resize(std::max<size_t>(bsR, 8)); hr_[0].resize(std::max<size_t>(brL, 8)); hr_[1].resize(std::max<size_t>(brR, 8));
    for (int a = 0; a < nArr; a++)
    {
        if (sLn) HIP_CHECK(hipMemcpyAsync(hs_[0].data() + sizeof(double) * (size_t)a * sLn, arrays[a] + sLb, sizeof(double) * (size_t)sLn, hipMemcpyDeviceToHost, stream));
        if (sRn) HIP_CHECK(hipMemcpyAsync(hs_[1].data() + sizeof(double) * (size_t)a * sRn, arrays[a] + sRb, sizeof(double) * (size_t)sRn, hipMemcpyDeviceToHost, stream));
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    int64_t got = 0;
    // leftward messages first (everyone sends left / receives from the right), then rightward ones - as in exchange()
    if (sr_(ctx_, left, hs_[0].data(), (int64_t)bsL, right, hr_[1].data(), (int64_t)brR, &got) != 0 || got != (int64_t)brR)
        throw std::runtime_error("slab coordinate exchange failed (leftward): the neighbours disagree about their boundary atoms");
    if (sr_(ctx_, right, hs_[1].data(), (int64_t)bsR, left, hr_[0].data(), (int64_t)brL, &got) != 0 || got != (int64_t)brL)
        throw std::runtime_error("slab coordinate exchange failed (rightward): the neighbours disagree about their boundary atoms");
    for (int a = 0; a < nArr; a++)
    {
        if (rLn) HIP_CHECK(hipMemcpyAsync(arrays[a] + rLb, hr_[0].data() + sizeof(double) * (size_t)a * rLn, sizeof(double) * (size_t)rLn, hipMemcpyHostToDevice, stream));
        if (rRn) HIP_CHECK(hipMemcpyAsync(arrays[a] + rRb, hr_[1].data() + sizeof(double) * (size_t)a * rRn, sizeof(double) * (size_t)rRn, hipMemcpyHostToDevice, stream));
    }
    HIP_CHECK(hipStreamSynchronize(stream));
}

void CallbackExchanger::exchange_counts(int left, int right, const int32_t* dToLeft, const int32_t* dToRight, int32_t* dFromLeft, int32_t* dFromRight, int n,
                                        hipStream_t stream)
{
    std::vector<int32_t> sl((size_t)n), sr((size_t)n), rl((size_t)n), rr((size_t)n);
    const size_t bytes = sizeof(int32_t) * (size_t)n;
    HIP_CHECK(hipMemcpyAsync(sl.data(), dToLeft, bytes, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(sr.data(), dToRight, bytes, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    int64_t got = 0;
    if (sr_(ctx_, left, sl.data(), (int64_t)bytes, right, rr.data(), (int64_t)bytes, &got) != 0 || got != (int64_t)bytes)
        throw std::runtime_error("slab count exchange callback failed (leftward)");
    if (sr_(ctx_, right, sr.data(), (int64_t)bytes, left, rl.data(), (int64_t)bytes, &got) != 0 || got != (int64_t)bytes)
        throw std::runtime_error("slab count exchange callback failed (rightward)");
    HIP_CHECK(hipMemcpyAsync(dFromLeft, rl.data(), bytes, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(dFromRight, rr.data(), bytes, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

void CallbackExchanger::allreduce_sum(double* host, int n, hipStream_t)
{
    if (ar_(ctx_, host, n) != 0) throw std::runtime_error("slab allreduce callback failed");
}

void CallbackExchanger::allreduce_device(double* dev, int n, hipStream_t stream)
{
    std::vector<double> h((size_t)n);
    HIP_CHECK(hipMemcpyAsync(h.data(), dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (ar_(ctx_, h.data(), n) != 0) throw std::runtime_error("slab allreduce callback failed");
    HIP_CHECK(hipMemcpyAsync(dev, h.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

// ---------------------------------------------------------------------------------------------------
// loopback transport (measurement aid)
// ---------------------------------------------------------------------------------------------------
__global__ void k_loopback_shift(char* msg, size_t migOff, size_t haloOff, size_t migStride, size_t haloStride, double shift, double L)
{
    const int32_t* hdr = (const int32_t*)msg;
    const int nMig = hdr[0], nHalo = hdr[1];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double* px = nullptr;
    if (t < nMig) px = (double*)(msg + migOff + (size_t)t * migStride);
    else if (t - nMig < nHalo) px = (double*)(msg + haloOff + (size_t)(t - nMig) * haloStride);
    if (px)
    {
        double x = *px + shift;
        if (x < 0) x += L; else if (x >= L) x -= L;
        *px = x;
    }
}

void LoopbackExchanger::exchange(int, int, const void* dSendLeft, const void* dSendRight, void* dFromLeft, void* dFromRight, size_t bytes,
                                 hipStream_t stream)
{
    HIP_CHECK(hipMemcpyAsync(dFromRight, dSendLeft, bytes, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(dFromLeft, dSendRight, bytes, hipMemcpyDeviceToDevice, stream));
    const int n = (int)((bytes - haloOff_) / haloStride_ + (haloOff_ - migOff_) / migStride_);
    hipLaunchKernelGGL(k_loopback_shift, dim3((n + 255) / 256), dim3(256), 0, stream, (char*)dFromRight, migOff_, haloOff_, migStride_, haloStride_, w_, L_);
    hipLaunchKernelGGL(k_loopback_shift, dim3((n + 255) / 256), dim3(256), 0, stream, (char*)dFromLeft, migOff_, haloOff_, migStride_, haloStride_, -w_, L_);
}

// one kernel for the whole loopback coordinate exchange (a grouped ncclSend/ncclRecv is one operation too): what goes left comes back from the
// right, one slab width further along x, and vice versa
struct LoopArrays { double* a[4]; };
__global__ void k_loopback_ranges(LoopArrays A, int nArr, int sLb, int sRb, int rLb, int rRb, int nToRight, int nToLeft, double w, double L)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < nArr; k++)
    {
        // Between two sorts coordinates stay unwrapped and a ghost moves continuously with its source (the recorded image codes and cell shifts assume
        // it): the copy is source + slab width + the multiple of L it had at the sort - recovered from the ghost's previous value, which is within an
        // atom's step of where it belongs.  (Re-wrapping here made a ghost jump by L when its source drifted across the seam: its pairs were missed.)
        if (t < nToRight)
        {   // sent leftward -> arrives as "from the right"
            double v = A.a[k][sLb + t];
            if (k == 0) { v += w; v += L * nearbyint((A.a[k][rRb + t] - v) / L); }
            A.a[k][rRb + t] = v;
        }
        if (t < nToLeft)
        {
            double v = A.a[k][sRb + t];
            if (k == 0) { v -= w; v += L * nearbyint((A.a[k][rLb + t] - v) / L); }
            A.a[k][rLb + t] = v;
        }
    }
}

void LoopbackExchanger::exchange_ranges(int, int, double* const* arrays, int nArr, int sLb, int sLn, int sRb, int sRn, int rLb, int rLn, int rRb, int rRn,
                                        hipStream_t stream)
{   // (sLn == rRn and sRn == rLn for a rank talking to itself)
    LoopArrays A{};
    if (nArr > 4) throw std::runtime_error("slab loopback transport: at most 4 arrays per coordinate exchange");
    for (int k = 0; k < nArr; k++) A.a[k] = arrays[k];
    const int nToRight = std::min(sLn, rRn), nToLeft = std::min(sRn, rLn), n = std::max(nToRight, nToLeft);
    if (n > 0) hipLaunchKernelGGL(k_loopback_ranges, dim3((n + 255) / 256), dim3(256), 0, stream, A, nArr, sLb, sRb, rLb, rRb, nToRight, nToLeft, w_, L_);
}

void LoopbackExchanger::exchange_counts(int, int, const int32_t* dToLeft, const int32_t* dToRight, int32_t* dFromLeft, int32_t* dFromRight, int n, hipStream_t stream)
{   // what goes left comes back as the right neighbour's message and vice versa
    HIP_CHECK(hipMemcpyAsync(dFromRight, dToLeft, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(dFromLeft, dToRight, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream));
}

// ---------------------------------------------------------------------------------------------------
// RCCL transport
// ---------------------------------------------------------------------------------------------------
namespace {
struct RcclApi
{
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl()
{
    static RcclApi api;
    if (api.lib) return api;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.lib) break; }
    if (!api.lib) throw std::runtime_error(std::string("cannot load RCCL: ") + dlerror());
    auto sym = [&](const char* s) { void* p = dlsym(api.lib, s); if (!p) throw std::runtime_error(std::string("RCCL symbol missing: ") + s); return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    return api;
}

void check_nccl(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL error in ") + what + ": " + rccl().GetErrorString(r));
}
}  // namespace

int RcclExchanger::id_bytes() { return (int)sizeof(ncclUniqueId); }
void RcclExchanger::make_id(void* out)
{
    ncclUniqueId id;
    check_nccl(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, &id, sizeof(id));
}

RcclExchanger::RcclExchanger(int rank, int nranks, const void* id_bytes)
{
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c;
    check_nccl(rccl().CommInitRank(&c, nranks, id, rank), "ncclCommInitRank");
    comm_ = c;
    HIP_CHECK(hipMalloc((void**)&dScratch_, sizeof(double) * 256));
}

RcclExchanger::~RcclExchanger()
{
    if (dScratch_) (void)hipFree(dScratch_);
    if (comm_) rccl().CommDestroy((ncclComm_t)comm_);
}

int RcclExchanger::comm_ranks() const
{
    int n = 0;
    check_nccl(rccl().CommCount((ncclComm_t)comm_, &n), "ncclCommCount");
    return n;
}

void RcclExchanger::exchange(int left, int right, const void* dSendLeft, const void* dSendRight, void* dFromLeft, void* dFromRight,
                             size_t bytes, hipStream_t stream)
{
    RcclApi& a = rccl();
    ncclComm_t c = (ncclComm_t)comm_;
    check_nccl(a.GroupStart(), "ncclGroupStart");
    // with two ranks both neighbours are the same peer: sends and receives are matched in issue order,
    // so post them in the same order on both sides (leftward message first)
    check_nccl(a.Send(dSendLeft, bytes, ncclChar, left, c, stream), "ncclSend(left)");
    check_nccl(a.Recv(dFromRight, bytes, ncclChar, right, c, stream), "ncclRecv(right)");
    check_nccl(a.Send(dSendRight, bytes, ncclChar, right, c, stream), "ncclSend(right)");
    check_nccl(a.Recv(dFromLeft, bytes, ncclChar, left, c, stream), "ncclRecv(left)");
    check_nccl(a.GroupEnd(), "ncclGroupEnd");
}

void RcclExchanger::exchange_ranges(int left, int right, double* const* arrays, int nArr, int sLb, int sLn, int sRb, int sRn, int rLb, int rLn, int rRb,
                                    int rRn, hipStream_t stream)
{
    RcclApi& a = rccl();
    ncclComm_t c = (ncclComm_t)comm_;
    check_nccl(a.GroupStart(), "ncclGroupStart");
    for (int k = 0; k < nArr; k++)
    {   // same pairing order as exchange(): with two ranks both neighbours are one peer and messages match in issue order
        check_nccl(a.Send(arrays[k] + sLb, (size_t)sLn, ncclDouble, left, c, stream), "ncclSend(left)");
        check_nccl(a.Recv(arrays[k] + rRb, (size_t)rRn, ncclDouble, right, c, stream), "ncclRecv(right)");
        check_nccl(a.Send(arrays[k] + sRb, (size_t)sRn, ncclDouble, right, c, stream), "ncclSend(right)");
        check_nccl(a.Recv(arrays[k] + rLb, (size_t)rLn, ncclDouble, left, c, stream), "ncclRecv(left)");
    }
    check_nccl(a.GroupEnd(), "ncclGroupEnd");
}

void RcclExchanger::exchange_counts(int left, int right, const int32_t* dToLeft, const int32_t* dToRight, int32_t* dFromLeft, int32_t* dFromRight, int n,
                                    hipStream_t stream)
{   // fixed size on every rank, so this exchange cannot itself be mismatched; same pairing order as exchange()
    RcclApi& a = rccl();
    ncclComm_t c = (ncclComm_t)comm_;
    check_nccl(a.GroupStart(), "ncclGroupStart");
    check_nccl(a.Send(dToLeft, (size_t)n, ncclInt32, left, c, stream), "ncclSend(left)");
    check_nccl(a.Recv(dFromRight, (size_t)n, ncclInt32, right, c, stream), "ncclRecv(right)");
    check_nccl(a.Send(dToRight, (size_t)n, ncclInt32, right, c, stream), "ncclSend(right)");
    check_nccl(a.Recv(dFromLeft, (size_t)n, ncclInt32, left, c, stream), "ncclRecv(left)");
    check_nccl(a.GroupEnd(), "ncclGroupEnd");
}

void RcclExchanger::allreduce_sum(double* host, int n, hipStream_t stream)
{
    if (n > 256) throw std::runtime_error("allreduce_sum: too many values");
    HIP_CHECK(hipMemcpyAsync(dScratch_, host, sizeof(double) * n, hipMemcpyHostToDevice, stream));
    check_nccl(rccl().AllReduce(dScratch_, dScratch_, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)comm_, stream), "ncclAllReduce");
    HIP_CHECK(hipMemcpyAsync(host, dScratch_, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

void RcclExchanger::allreduce_device(double* dev, int n, hipStream_t stream)
{
    check_nccl(rccl().AllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)comm_, stream), "ncclAllReduce");
}

void RcclExchanger::selftest(int device)
{
    HIP_CHECK(hipSetDevice(device));
    char id[256];
    if (id_bytes() > (int)sizeof(id)) throw std::runtime_error("RCCL unique id larger than expected");
    make_id(id);
    RcclExchanger x(0, 1, id);
    hipStream_t st;
    HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t bytes = 1 << 20;
    char* d[4];
    for (auto& p : d) HIP_CHECK(hipMalloc((void**)&p, bytes));
    std::vector<char> a(bytes), b(bytes), ra(bytes), rb(bytes);
    for (size_t i = 0; i < bytes; i++) { a[i] = (char)(i * 7 + 1); b[i] = (char)(i * 13 + 5); }
    HIP_CHECK(hipMemcpyAsync(d[0], a.data(), bytes, hipMemcpyHostToDevice, st));    // leftward message
    HIP_CHECK(hipMemcpyAsync(d[1], b.data(), bytes, hipMemcpyHostToDevice, st));    // rightward message
    HIP_CHECK(hipMemsetAsync(d[2], 0, bytes, st));
    HIP_CHECK(hipMemsetAsync(d[3], 0, bytes, st));
    x.exchange(0, 0, d[0], d[1], d[2], d[3], bytes, st);                              // fromLeft = d[2], fromRight = d[3]
    HIP_CHECK(hipMemcpyAsync(ra.data(), d[3], bytes, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(rb.data(), d[2], bytes, hipMemcpyDeviceToHost, st));
    double h[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    double* dd;
    HIP_CHECK(hipMalloc((void**)&dd, sizeof(h)));
    HIP_CHECK(hipMemcpyAsync(dd, h, sizeof(h), hipMemcpyHostToDevice, st));
    x.allreduce_device(dd, 8, st);
    double back[8];
    HIP_CHECK(hipMemcpyAsync(back, dd, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    double hs[3] = {0.5, -2.0, 1e300};
    x.allreduce_sum(hs, 3, st);
    // the two exchanges of a lazy run's plain steps, with themselves as both neighbours: the fixed-size count messages and the coordinate ranges
    // (three arrays, different counts leftward and rightward) - same grouped ncclSend / ncclRecv calls as between real ranks
    int32_t hc[8] = {11, 22, 33, 44, 0, 0, 0, 0}, hcBack[8];
    int32_t* dc;
    HIP_CHECK(hipMalloc((void**)&dc, sizeof(hc)));
    HIP_CHECK(hipMemcpyAsync(dc, hc, sizeof(hc), hipMemcpyHostToDevice, st));
    x.exchange_counts(0, 0, dc, dc + 2, dc + 6, dc + 4, 2, st);          // to left {11, 22}, to right {33, 44}; from right -> [4..5], from left -> [6..7]
    HIP_CHECK(hipMemcpyAsync(hcBack, dc, sizeof(hc), hipMemcpyDeviceToHost, st));
    const int nR = 64;
    std::vector<double> ha(3 * nR), haBack(3 * nR);
    for (int i = 0; i < 3 * nR; i++) ha[i] = 1000.0 * (i / nR) + (i % nR);
    double* da;
    HIP_CHECK(hipMalloc((void**)&da, sizeof(double) * 3 * nR));
    HIP_CHECK(hipMemcpyAsync(da, ha.data(), sizeof(double) * 3 * nR, hipMemcpyHostToDevice, st));
    double* arrs[3] = {da, da + nR, da + 2 * nR};
    // layout of every array: [0, 5) left ghosts | [5, 12) goes left | ... | [40, 50) goes right | [50, 57) right ghosts: 7 elements arrive from the right, 5 from the left
    x.exchange_ranges(0, 0, arrs, 3, 5, 7, 45, 5, 0, 5, 50, 7, st);
    HIP_CHECK(hipMemcpyAsync(haBack.data(), da, sizeof(double) * 3 * nR, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(dc); (void)hipFree(da);
    bool rangesOk = hcBack[4] == 11 && hcBack[5] == 22 && hcBack[6] == 33 && hcBack[7] == 44;
    for (int k = 0; k < 3 && rangesOk; k++)
    {
        for (int i = 0; i < 7; i++) rangesOk = rangesOk && haBack[k * nR + 50 + i] == ha[k * nR + 5 + i];       // what went left arrived as "from the right"
        for (int i = 0; i < 5; i++) rangesOk = rangesOk && haBack[k * nR + i] == ha[k * nR + 45 + i];           // what went right arrived as "from the left"
    }
    for (auto p : d) (void)hipFree(p);
    (void)hipFree(dd);
    (void)hipStreamDestroy(st);
    // what a rank sends leftward must arrive as its neighbour's "from the right" message, and vice versa
    if (ra != a || rb != b) throw std::runtime_error("RCCL self-test: ring exchange delivered the wrong payload");
    for (int k = 0; k < 8; k++) if (back[k] != h[k]) throw std::runtime_error("RCCL self-test: device all-reduce mismatch");
    if (hs[0] != 0.5 || hs[1] != -2.0 || hs[2] != 1e300) throw std::runtime_error("RCCL self-test: host all-reduce mismatch");
    if (!rangesOk) throw std::runtime_error("RCCL self-test: count / coordinate-range exchange delivered the wrong payload");
}

}  // namespace aztot
