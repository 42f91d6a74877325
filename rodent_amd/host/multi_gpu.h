// multi_gpu.h -- the GPUs of one node behind the C++ hosts (`rodent --ngpu K`, `bench_traversal -ngpu K`): one host thread per
// device for the compute, and the ONE collective of the path -- a gather of disjoint parts (row bands of the film,
// ranges of the Hit1 array) to the root device -- as grouped RCCL point-to-point calls: the root posts one ncclRecv per peer
// straight into that peer's place in ITS buffer, every peer one ncclSend of its own part.  Each byte crosses one xGMI link
// once (7 links x ~153 GB/s per GPU, point to point: the peers' sends do not share a link), nothing is padded, nobody but the
// root receives anything.  With one device nothing is initialised and nothing is sent.
// (The Python hosts do the same through torch.distributed: rodent_amd/parallel.py gather_parts_to_root.)
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "partition.h"

namespace rodent {

class DeviceGroup {
public:
    ~DeviceGroup() {
        for (size_t k = 0; k < comms_.size(); k++) ncclCommDestroy(comms_[k]);
        for (size_t k = 0; k < streams_.size(); k++) { (void)hipSetDevice(devs_[k]); (void)hipStreamDestroy(streams_[k]); }
    }
    // devices first .. first + count - 1; false (with a message in *err) if the node does not have them or RCCL does not come up
    bool init(int first, int count, std::string* err) {
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess || first < 0 || count < 1 || first + count > have) {
            *err = "the node has " + std::to_string(have) + " GPU device(s), " + std::to_string(count) + " from device " + std::to_string(first) + " on were asked for";
            return false;
        }
        for (int k = 0; k < count; k++) devs_.push_back(first + k);
        if (count == 1) return true;
        comms_.resize(count);
        const ncclResult_t r = ncclCommInitAll(comms_.data(), count, devs_.data());
        if (r != ncclSuccess) { comms_.clear(); *err = std::string("ncclCommInitAll: ") + ncclGetErrorString(r); return false; }
        streams_.resize(count);
        for (int k = 0; k < count; k++) {
            if (hipSetDevice(devs_[k]) != hipSuccess || hipStreamCreateWithFlags(&streams_[k], hipStreamNonBlocking) != hipSuccess) { *err = "cannot create a stream"; return false; }
        }
        return true;
    }
    int size() const { return (int)devs_.size(); }
    int device(int rank) const { return devs_[rank]; }

    // work(rank) on one host thread per device, all at once; returns when every one has returned
    void run(const std::function<void(int)>& work) const {
        if (size() == 1) { work(0); return; }
        std::vector<std::thread> threads;
        for (int k = 0; k < size(); k++) threads.emplace_back(work, k);
        for (auto& t : threads) t.join();
    }

    // Rank r's `bytes[r]` bytes at `src[r]` (on device r) go to `dst[r]` on the root device (rank 0); the root's own part is in
    // place already.  One group of sends / receives, then every stream is waited for.  Returns the seconds it took, < 0 on error.
    double gather_to_root(const std::vector<const void*>& src, const std::vector<void*>& dst, const std::vector<size_t>& bytes, std::string* err) const {
        if (size() == 1) return 0.0;
        const auto t0 = std::chrono::steady_clock::now();
        ncclResult_t r = ncclGroupStart();
        for (int k = 1; k < size() && r == ncclSuccess; k++) {
            if (!bytes[k]) continue;
            r = ncclSend(src[k], bytes[k], ncclChar, 0, comms_[k], streams_[k]);
            if (r == ncclSuccess) r = ncclRecv(dst[k], bytes[k], ncclChar, k, comms_[0], streams_[0]);
        }
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { *err = std::string("RCCL gather: ") + ncclGetErrorString(r); return -1.0; }
        for (int k = 0; k < size(); k++)
            if (hipSetDevice(devs_[k]) != hipSuccess || hipStreamSynchronize(streams_[k]) != hipSuccess) { *err = "RCCL gather: stream synchronisation failed"; return -1.0; }
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

private:
    std::vector<int> devs_;
    std::vector<ncclComm_t> comms_;
    std::vector<hipStream_t> streams_;
};

} // namespace rodent
