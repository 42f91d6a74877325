// multi_gpu.h -- the GPUs of one node behind the C++ hosts (`rodent --ngpu K`, `bench_traversal -ngpu K`): one host thread per
// device for the compute (kept alive between frames: a frame of a small scene is shorter than creating K threads), and the ONE
// collective of the path -- a gather of disjoint parts (row tiles of the film, ranges of the Hit1 array) to the root
// device -- as grouped RCCL point-to-point calls: the root posts one ncclRecv per part straight into its place in ITS buffer,
// the part's owner one ncclSend.  Each byte crosses one xGMI link
// once (7 links x ~153 GB/s per GPU, point to point: the peers' sends do not share a link), nothing is padded, nobody but the
// root receives anything.  With one device nothing is initialised and nothing is sent.
// The K > 1 path has never run on more than one GPU before the driver's first multi-GPU bench, so it must not fail silently there (VERDICT
// r4 item 7): if RCCL does not come up the gather falls back to one hipMemcpyPeerAsync per piece, with a warning; describe() says which
// transport is in use, which RCCL version, and how many ranks the communicator reports.  RODENT_SHARE_GPUS=1 (tests) maps the K ranks onto
// the devices the box has (rank r -> device first + r mod have): RCCL cannot put two ranks on one device, so this exercises threads +
// partition + the fallback gather on one GPU; RODENT_FORCE_RCCL_INIT_FAILURE=1 injects the failure on a box where RCCL would work.
// (The Python hosts do the same through torch.distributed: rodent_amd/parallel.py gather_parts_to_root.)
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "partition.h"

namespace rodent {

class DeviceGroup {
public:
    ~DeviceGroup() {
        {
            std::lock_guard<std::mutex> lock(m_);
            quit_ = true;
        }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
        for (size_t k = 0; k < comms_.size(); k++) ncclCommDestroy(comms_[k]);
        for (size_t k = 0; k < streams_.size(); k++) { (void)hipSetDevice(devs_[k]); (void)hipStreamDestroy(streams_[k]); }
    }
    // devices first .. first + count - 1; false (with a message in *err) if the node does not have them or RCCL does not come up
    bool init(int first, int count, std::string* err) {
        int have = 0;
        const char* share = getenv("RODENT_SHARE_GPUS");
        const bool shared = share && atoi(share) != 0;
        if (hipGetDeviceCount(&have) != hipSuccess || first < 0 || count < 1 || have < 1 || (!shared && first + count > have)) {
            *err = "the node has " + std::to_string(have) + " GPU device(s), " + std::to_string(count) + " from device "
                + std::to_string(first) + " on were asked for";
            return false;
        }
        for (int k = 0; k < count; k++) devs_.push_back(shared ? (first + k) % have : first + k);
        if (count == 1) return true;
        streams_.resize(count);
        for (int k = 0; k < count; k++) {
            if (hipSetDevice(devs_[k]) != hipSuccess || hipStreamCreateWithFlags(&streams_[k], hipStreamNonBlocking) != hipSuccess) {
                *err = "cannot create a stream"; return false; }
        }
        int version = 0;
        if (ncclGetVersion(&version) == ncclSuccess) rccl_version_ = version;
        const char* inject = getenv("RODENT_FORCE_RCCL_INIT_FAILURE");
        const bool injected = inject && atoi(inject) != 0;
        std::string init_error;
        if (rccl_unused_reason(count, have, shared, injected, "").empty()) {          // nothing rules RCCL out beforehand: bring it up
            comms_.resize(count);
            const ncclResult_t r = ncclCommInitAll(comms_.data(), count, devs_.data());
            if (r != ncclSuccess) { comms_.clear(); init_error = ncclGetErrorString(r); }
            else if (ncclCommCount(comms_[0], &comm_ranks_) != ncclSuccess) comm_ranks_ = -1;
        }
        const std::string why = rccl_unused_reason(count, have, shared, injected, init_error);
        if (!why.empty()) {
            fallback_ = why;
            std::cerr << "rodent: WARNING: RCCL is not used (" << why
                << "); the gather to the first device falls back to one hipMemcpyPeerAsync per piece" << std::endl;
            // peer access where the hardware offers it (the copies work without, through the host)
            for (int k = 1; k < count; k++) {
                int can = 0;
                if (devs_[k] != devs_[0] && hipDeviceCanAccessPeer(&can, devs_[0], devs_[k]) == hipSuccess && can) {
                    (void)hipSetDevice(devs_[0]); (void)hipDeviceEnablePeerAccess(devs_[k], 0); (void)hipGetLastError(); }
            }
        }
        return true;
    }
    // one line for the tools' stdout: which transport the gather uses and what the communicator reports
    std::string describe() const {
        if (size() == 1) return "one device, no collective";
        const std::string v = rccl_version_
            ? std::to_string(rccl_version_ / 10000) + "." + std::to_string(rccl_version_ / 100 % 100) + "."
            + std::to_string(rccl_version_ % 100) : std::string("unknown");
        if (!fallback_.empty()) return "hipMemcpyPeerAsync per piece (RCCL " + v + " not used: " + fallback_ + ")";
        return "RCCL " + v + ", communicator of " + std::to_string(comm_ranks_) + " rank(s), grouped ncclSend / ncclRecv";
    }
    bool uses_rccl() const { return size() > 1 && fallback_.empty(); }
    int size() const { return (int)devs_.size(); }
    int device(int rank) const { return devs_[rank]; }

    // work(rank) on the device's own host thread, all at once; returns when every one has returned.  The threads are created at the
    // first call and wait for the next one (a generation counter under one mutex: K is 8 at most).
    void run(const std::function<void(int)>& work) {
        if (size() == 1) { work(0); return; }
        std::unique_lock<std::mutex> lock(m_);
        if (workers_.empty())
            for (int k = 0; k < size(); k++) workers_.emplace_back([this, k] { worker(k); });
        work_ = &work; pending_ = size(); generation_++;
        wake_.notify_all();
        done_.wait(lock, [this] { return pending_ == 0; });
        work_ = nullptr;
    }

    // One part of a gather: `bytes` bytes at `src` on rank `rank`'s device go to `dst` on the root device (rank 0).
    struct Piece { int rank; const void* src; void* dst; size_t bytes; };
    // All pieces in ONE group of sends / receives (the root's own parts are in place already: pieces of rank 0 are skipped), then every
    // stream is waited for.  Returns the seconds it took, < 0 on error.
    double gather_to_root(const std::vector<Piece>& pieces, std::string* err) const {
        if (size() == 1) return 0.0;
        const auto t0 = std::chrono::steady_clock::now();
        if (!fallback_.empty()) {
            for (const Piece& p : pieces) {
                if (p.rank == 0 || !p.bytes) continue;
                if (hipSetDevice(devs_[p.rank]) != hipSuccess
                    || hipMemcpyPeerAsync(p.dst, devs_[0], p.src, devs_[p.rank], p.bytes, streams_[p.rank]) != hipSuccess) {
                    *err = "peer-copy gather: hipMemcpyPeerAsync failed"; return -1.0; }
            }
            for (int k = 0; k < size(); k++)
                if (hipSetDevice(devs_[k]) != hipSuccess || hipStreamSynchronize(streams_[k]) != hipSuccess) {
                    *err = "peer-copy gather: stream synchronisation failed"; return -1.0; }
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        ncclResult_t r = ncclGroupStart();
        for (size_t i = 0; i < pieces.size() && r == ncclSuccess; i++) {
            const Piece& p = pieces[i];
            if (p.rank == 0 || !p.bytes) continue;
            r = ncclSend(p.src, p.bytes, ncclChar, 0, comms_[p.rank], streams_[p.rank]);
            if (r == ncclSuccess) r = ncclRecv(p.dst, p.bytes, ncclChar, p.rank, comms_[0], streams_[0]);
        }
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { *err = std::string("RCCL gather: ") + ncclGetErrorString(r); return -1.0; }
        for (int k = 0; k < size(); k++)
            if (hipSetDevice(devs_[k]) != hipSuccess || hipStreamSynchronize(streams_[k]) != hipSuccess) {
                *err = "RCCL gather: stream synchronisation failed"; return -1.0; }
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

private:
    void worker(int rank) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)>* work = nullptr;
            {
                std::unique_lock<std::mutex> lock(m_);
                wake_.wait(lock, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_; work = work_;
            }
            (*work)(rank);
            {
                std::lock_guard<std::mutex> lock(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(int)>* work_ = nullptr;
    unsigned long long generation_ = 0;
    int pending_ = 0;
    bool quit_ = false;
    std::vector<int> devs_;
    std::vector<ncclComm_t> comms_;
    std::vector<hipStream_t> streams_;
    std::string fallback_;                      // why RCCL is not used (empty: it is)
    int rccl_version_ = 0, comm_ranks_ = 0;
};

} // namespace rodent
