// gpx_host.hpp -- the host-side concurrency machinery of libgpx.so, free of any HIP type so that it can be built
// and run under ThreadSanitizer / AddressSanitizer on a CPU-only machine (tests/cpp/host_concurrency.cpp links it with
// a stub device backend):
//   * DeviceBackend   : the four device calls this machinery needs (current device, set device, allocate, free);
//                       libgpx.so installs the HIP implementation (gpx_api.hip), the test a stub with injectable faults
//   * BigPool         : per-process pool of large device buffers (see gpx_model.hpp for why it exists)
//   * PerDeviceOnce   : run-once-per-device-ordinal flag for kernel attributes
//   * FlatCombiner<R> : concurrent small requests on one model are merged into one device batch by whichever caller
//                       finds no leader (the node issues one evaluate per grid point from hundreds of threads,
//                       reference src/gp_node.cpp:1027-1038)
//   * eigen_pivot_order : Eigen 3.2 LDLT's pivot sequence from the diagonal (reference gp_regressor.hpp:161-162)
//   * Switches        : the developer / test switches of DESIGN section 10, parsed from the environment ONCE
#pragma once
#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <string>
#include <vector>

namespace gpxh {

constexpr int MAX_DEVICES = 64;

struct DeviceBackend {
    int (*get_device)();                        // current ordinal, or -1 when it cannot be told
    void (*set_device)(int dev);
    int (*dev_malloc)(void **p, size_t bytes);  // 0 = ok, 1 = out of memory, 2 = any other failure
    void (*dev_free)(void *p);
};
void set_device_backend(const DeviceBackend *b);  // must be called before any other entry of this header
const DeviceBackend *device_backend();

// ---- pool of large device buffers ------------------------------------------------------------------------------------
// (4 KiB since round 3, 64 MiB before: a model of a few hundred points is ~15 buffers of 10 KiB .. 2 MiB, and their
// hipMalloc / hipFree calls were half of its create + destroy time; at most BIG_POOL_MAX_PARKED buffers are kept)
constexpr size_t BIG_POOL_MIN = (size_t)4 << 10;
constexpr size_t BIG_POOL_MAX_PARKED = 512;
class BigPool {
public:
    explicit BigPool(size_t cap_bytes) : cap_(cap_bytes) {}
    // 0 = ok, 1 = out of memory (after emptying the pool and retrying once), 2 = other failure
    int alloc(void **p, size_t bytes);
    // The caller has already waited for the work that used p.  Buffers that came from alloc() with >= BIG_POOL_MIN bytes
    // are parked while the cap allows, everything else goes back to the device.
    void release(void *p);
    void trim();  // free every parked buffer
    size_t parked_bytes();
    size_t live_buffers();

private:
    struct Buf {
        void *p;
        size_t bytes;
        int dev;
    };
    std::mutex mtx_;
    std::vector<Buf> free_, live_;
    size_t parked_ = 0;
    const size_t cap_;
};

// ---- once per device ordinal --------------------------------------------------------------------------------------------
struct PerDeviceOnce {
    std::once_flag flag[MAX_DEVICES];
    template <typename F>
    void run(F &&f)
    {
        const DeviceBackend *b = device_backend();
        const int dev = b ? b->get_device() : -1;
        if (dev < 0 || dev >= MAX_DEVICES) {
            f();  // unknown ordinal: do it every time (the attribute calls are idempotent)
            return;
        }
        std::call_once(flag[dev], f);
    }
};

// ---- flat combining ------------------------------------------------------------------------------------------------------
// R needs the members   int rc;  bool done;  std::string err;
// submit(req, run): the calling thread either becomes the leader -- takes every pending request, calls
// run(batch, err) once for all of them (rc != 0: err holds the message) and hands the outcome to each -- or waits until a
// leader has served it.  Requests are served exactly once; a leader serves at least its own request.
template <typename R>
class FlatCombiner {
public:
    template <typename Run>
    int submit(R &req, Run &&run)
    {
        std::unique_lock<std::mutex> lk(mtx_);
        pending_.push_back(&req);
        while (!req.done) {
            if (!leader_active_) {
                leader_active_ = true;
                std::vector<R *> batch;
                batch.swap(pending_);
                lk.unlock();
                std::string err;
                const int brc = run(batch, err);
                lk.lock();
                for (R *p : batch) {
                    p->rc = brc;
                    if (brc)
                        p->err = err;
                    p->done = true;
                }
                leader_active_ = false;
                cv_.notify_all();
            } else {
                cv_.wait(lk);
            }
        }
        return req.rc;
    }

private:
    std::mutex mtx_;
    std::condition_variable cv_;
    std::vector<R *> pending_;
    bool leader_active_ = false;
};

// Eigen 3.2 LDLT pivot rule restated: at step k pick the FIRST largest |diagonal| among the not-yet-eliminated rows and
// swap it to k.  The left-looking algorithm never updates the trailing diagonal before it is chosen, so the sequence
// depends on diag(K) only.  perm: internal position -> caller index.
void eigen_pivot_order(const std::vector<double> &diag, std::vector<int> &perm);

// ---- developer / test switches -----------------------------------------------------------------------
// Every switch the library honours, read from the environment once per process (first use) and again only on
// gpx_debug_reload() (include/gpx.h) -- no getenv on any call path (round 5 had 32 sites, several per evaluate).  -1 = unset:
// the default, which is the product; each twin a switch selects is held to the default by a test.
struct Switches {
    long pool_mb = -1;         // GPX_POOL_MB         cap of the buffer pool (read when the pool is first used)
    int train_f64_max = -1;    // GPX_TRAIN_F64_MAX   F32 / F32_SPLIT models of up to this many padded rows train in fp64 (default 2048)
    int dataflow = -1;         // GPX_DATAFLOW        0: the launch chain at every size (twin of the one-launch creates);
                               //                     64 | 128: that tile form of the dataflow factorisation at every size above the small models
    long wait_budget_us = -1;  // GPX_WAIT_BUDGET_US  time after which a wait inside a dataflow launch / the one-launch substitution gives up
                               //                     (tests: 0 forces the give-up and with it the fallback paths)
    int update_append = -1;    // GPX_UPDATE_APPEND   0: update() always rebuilds (twin of the rank-n append)
    int dgp_append = -1;       // GPX_DGP_APPEND      0: gpx_dgp_add rebuilds on the union
    int var_cols = -1;         // GPX_VAR_COLS        0: small models through the general 128 x 128 tiles
    int var_cols16 = -1;       // GPX_VAR_COLS16      0: small split-mode models on the fp32 small-model kernel
    int var_cols64 = -1;       // GPX_VAR_COLS64      0: small fp64 models through the general path; 1: the one-wave-per-SIMD form of the small kernel
    int var_tile = -1;         // GPX_VAR_TILE        3: the LDS-staged four-wave tile instead of the one-wave tile
    int var_fit = -1;          // GPX_VAR_FIT         0: plain kernel values in the fp32 contraction (no per-query fit)
    int no_promote = -1;       // GPX_NO_PROMOTE      1: an indefinite fp32-mode model is rounded to fp32 all the same
    int small_eval = -1;       // GPX_SMALL_EVAL      0: few-query calls through the general path
    int project_fused = -1;    // GPX_PROJECT_FUSED   0: gpx_model_project as a host loop over evaluate calls
    int inv64 = -1;            // GPX_INV64           0: the inverse factor of fp32 models assembled in fp32
};
const Switches &switches();
void switches_reload();  // re-read the environment (tests); not while other threads are inside the library

}  // namespace gpxh
