// Transport of the slab decomposition: one fixed-size message to each x-neighbour per step.
//  * RcclExchanger    - ncclSend/ncclRecv on the engine's stream (RCCL over xGMI; point-to-point with the
//                       two ring neighbours only, so every transfer rides its own xGMI link).
//  * CallbackExchanger- host-staged through a caller-supplied function (gloo in the tests); correctness only.
// RCCL is loaded with dlopen at first use, so the library itself has no link-time dependency on it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/aztot.h"

namespace aztot {

class Exchanger
{
public:
    virtual ~Exchanger() {}
    // send `bytes` from dSendLeft to `left` and from dSendRight to `right`; receive the neighbours' messages
    // (same size) into dFromLeft / dFromRight.  Ordered on `stream`.
    virtual void exchange(int left, int right, const void* dSendLeft, const void* dSendRight, void* dFromLeft, void* dFromRight,
                          size_t bytes, hipStream_t stream) = 0;
    // Lazy re-sort, plain step: both neighbours hold the same atoms in the same order as after the last full exchange, so only coordinates
    // travel, straight between the per-atom arrays.  For each of the nArr arrays: elements [sLb, sLb + sLn) go to `left`, [sRb, sRb + sRn) to
    // `right`; [rLb, rLb + rLn) arrive from `left`, [rRb, rRb + rRn) from `right` (the sender's counts equal the receiver's by construction)
    virtual void exchange_ranges(int left, int right, double* const* arrays, int nArr, int sLb, int sLn, int sRb, int sRn, int rLb, int rLn, int rRb,
                                 int rRn, hipStream_t stream) = 0;
    // n 32-bit words to each neighbour and from each neighbour, between device buffers, ordered on `stream` (once per rebuild step: the ranks tell each
    // other how many boundary atoms they will send on the plain steps, so that a disagreement is an error before a mismatched send / receive is posted)
    virtual void exchange_counts(int left, int right, const int32_t* dToLeft, const int32_t* dToRight, int32_t* dFromLeft, int32_t* dFromRight, int n,
                                 hipStream_t stream) = 0;
    // element-wise sum over ranks of n doubles held on the HOST (statistics; not on the per-step path)
    virtual void allreduce_sum(double* host, int n, hipStream_t stream) = 0;
    // element-wise sum over ranks of n doubles held on the DEVICE, ordered on `stream` (Ewald structure factors: a real
    // exchange step of the path, once per force evaluation)
    virtual void allreduce_device(double* dev, int n, hipStream_t stream) = 0;
    virtual bool device_side() const = 0;
    // ranks of the RCCL communicator that carries the halo (ncclCommCount); 0 for every other transport
    virtual int comm_ranks() const { return 0; }
};

class CallbackExchanger : public Exchanger
{
public:
    CallbackExchanger(aztot_sendrecv_fn sr, aztot_allreduce_fn ar, void* ctx) : sr_(sr), ar_(ar), ctx_(ctx) {}
    void exchange(int left, int right, const void* dSendLeft, const void* dSendRight, void* dFromLeft, void* dFromRight, size_t bytes,
                  hipStream_t stream) override;
    void exchange_ranges(int left, int right, double* const* arrays, int nArr, int sLb, int sLn, int sRb, int sRn, int rLb, int rLn, int rRb, int rRn,
                         hipStream_t stream) override;
    void exchange_counts(int left, int right, const int32_t* dToLeft, const int32_t* dToRight, int32_t* dFromLeft, int32_t* dFromRight, int n,
                         hipStream_t stream) override;
    void allreduce_sum(double* host, int n, hipStream_t stream) override;
    void allreduce_device(double* dev, int n, hipStream_t stream) override;
    bool device_side() const override { return false; }

private:
    aztot_sendrecv_fn sr_;
    aztot_allreduce_fn ar_;
    void* ctx_;
    std::vector<char> hs_[2], hr_[2];
};

// Measurement aid: ONE rank of an N-rank decomposition talks to itself.  What it sends leftward comes back as if the right
// neighbour had sent it (x shifted by one slab width) and vice versa, so the rank sees the message sizes, ghost counts and
// kernel shapes of a real N-rank run on a single GPU.  The physics is that of a system made of N copies of this slab.
class LoopbackExchanger : public Exchanger
{
public:
    LoopbackExchanger(double slabWidth, double boxLength, size_t migOffset, size_t haloOffset, size_t migStride, size_t haloStride)
        : w_(slabWidth), L_(boxLength), migOff_(migOffset), haloOff_(haloOffset), migStride_(migStride), haloStride_(haloStride) {}
    void exchange(int left, int right, const void* dSendLeft, const void* dSendRight, void* dFromLeft, void* dFromRight, size_t bytes,
                  hipStream_t stream) override;
    void exchange_ranges(int left, int right, double* const* arrays, int nArr, int sLb, int sLn, int sRb, int sRn, int rLb, int rLn, int rRb, int rRn,
                         hipStream_t stream) override;
    void exchange_counts(int left, int right, const int32_t* dToLeft, const int32_t* dToRight, int32_t* dFromLeft, int32_t* dFromRight, int n,
                         hipStream_t stream) override;
    void allreduce_sum(double*, int, hipStream_t) override {}
    void allreduce_device(double*, int, hipStream_t) override {}
    bool device_side() const override { return true; }

private:
    double w_, L_;
    size_t migOff_, haloOff_, migStride_, haloStride_;
};

class RcclExchanger : public Exchanger
{
public:
    RcclExchanger(int rank, int nranks, const void* id_bytes);
    ~RcclExchanger() override;
    void exchange(int left, int right, const void* dSendLeft, const void* dSendRight, void* dFromLeft, void* dFromRight, size_t bytes,
                  hipStream_t stream) override;
    void exchange_ranges(int left, int right, double* const* arrays, int nArr, int sLb, int sLn, int sRb, int sRn, int rLb, int rLn, int rRb, int rRn,
                         hipStream_t stream) override;
    void exchange_counts(int left, int right, const int32_t* dToLeft, const int32_t* dToRight, int32_t* dFromLeft, int32_t* dFromRight, int n,
                         hipStream_t stream) override;
    void allreduce_sum(double* host, int n, hipStream_t stream) override;
    void allreduce_device(double* dev, int n, hipStream_t stream) override;
    bool device_side() const override { return true; }
    int comm_ranks() const override;
    static int id_bytes();
    static void make_id(void* out);
    // one-rank communicator on `device`: the ring exchange with itself (both messages, same call order as on N GPUs) and both
    // all-reduce flavours through the real RCCL library; throws on any mismatch
    static void selftest(int device);

private:
    void* comm_ = nullptr;
    double* dScratch_ = nullptr;
};

}  // namespace aztot
