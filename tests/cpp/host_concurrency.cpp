// host_concurrency.cpp -- the host-side concurrency machinery of libgpx.so (csrc/gpx_host.{hpp,cpp}: flat combining of
// concurrent evaluate() calls, the pool of large device buffers, the per-device once flags, Eigen's pivot order) under
// ThreadSanitizer and AddressSanitizer + UBSan ON THE CPU, against a stub device backend with injectable allocation
// failures.  The caller contract it exercises is the reference node's: one std::thread per grid point calling evaluate
// on a shared model (src/gp_node.cpp:1027-1038: 29 x 29 = 841 threads per x-slice) while another thread rebuilds
// models (:1093-1094).  Built and run by tests/test_host.py; exits 0 and prints "host concurrency ok" on success.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>

#include "gpx_host.hpp"

using namespace gpxh;

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond);   \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

// ---- stub device: host memory, a per-thread "current device", every n-th allocation fails on request ------------------
namespace stub {
thread_local int current = 0;
std::atomic<long> allocs{0}, frees{0}, live_bytes{0};
std::atomic<int> fail_every{0};      // n > 0: every n-th dev_malloc reports out of memory
std::atomic<int> fail_always_oom{0};
std::mutex size_mtx;
std::map<void *, size_t> sizes;
int get_device() { return current; }
void set_device(int d) { current = d; }
int dev_malloc(void **p, size_t bytes)
{
    const long k = ++allocs;
    const int fe = fail_every.load();
    if (fail_always_oom.load() || (fe > 0 && k % fe == 0)) {
        *p = nullptr;
        return 1;
    }
    // the pool only tracks sizes: a "1 GiB device buffer" is 64 real bytes here
    *p = std::malloc(64);
    if (!*p)
        return 2;
    std::lock_guard<std::mutex> lk(size_mtx);
    sizes[*p] = bytes;
    live_bytes += (long)bytes;
    return 0;
}
void dev_free(void *p)
{
    if (!p)
        return;
    {
        std::lock_guard<std::mutex> lk(size_mtx);
        auto it = sizes.find(p);
        CHECK(it != sizes.end());  // double free / foreign pointer
        live_bytes -= (long)it->second;
        sizes.erase(it);
    }
    ++frees;
    std::free(p);
}
const DeviceBackend backend{get_device, set_device, dev_malloc, dev_free};
}  // namespace stub

// ---- 1. flat combining: 841 threads x single-point requests -----------------------------------------------------------
struct Req {
    double qx, qy, qz, f = 0;
    int rc = 0;
    bool done = false;
    std::string err;
};

static void test_flat_combining(int nthreads, int rounds)
{
    FlatCombiner<Req> comb;
    std::atomic<long> batches{0}, served{0}, max_batch{0};
    std::atomic<int> inside{0};
    auto run = [&](std::vector<Req *> &batch, std::string &err) -> int {
        // exactly one leader at a time (what serialises the device work of a model)
        CHECK(inside.fetch_add(1) == 0);
        ++batches;
        long mb = max_batch.load();
        while ((long)batch.size() > mb && !max_batch.compare_exchange_weak(mb, (long)batch.size())) {
        }
        int rc = 0;
        std::this_thread::sleep_for(std::chrono::microseconds(200));  // a device round trip: callers pile up meanwhile
        for (Req *r : batch) {
            r->f = r->qx + 2 * r->qy + 3 * r->qz;  // stands in for the device batch
            ++served;
            if (r->qx < 0)
                rc = -7, err = "injected device error";  // one bad request fails its whole batch, message included
        }
        inside.fetch_sub(1);
        return rc;
    };
    for (int round = 0; round < rounds; ++round) {
        std::vector<std::thread> th;
        std::vector<Req> reqs((size_t)nthreads);
        std::vector<int> rcs((size_t)nthreads, 12345);
        for (int i = 0; i < nthreads; ++i) {
            reqs[i].qx = (round == 1 && i == 17) ? -1.0 : i;
            reqs[i].qy = round;
            reqs[i].qz = 0.5 * i;
            th.emplace_back([&, i] { rcs[i] = comb.submit(reqs[i], run); });
        }
        for (auto &t : th)
            t.join();
        int failed = 0;
        for (int i = 0; i < nthreads; ++i) {
            CHECK(reqs[i].done);
            CHECK(rcs[i] == reqs[i].rc);
            CHECK(reqs[i].f == reqs[i].qx + 2 * reqs[i].qy + 3 * reqs[i].qz);  // served exactly once, by someone
            if (rcs[i]) {
                ++failed;
                CHECK(reqs[i].err == "injected device error");
            }
        }
        CHECK((round == 1) == (failed > 0));
    }
    CHECK(served.load() == (long)nthreads * rounds);
    CHECK(nthreads < 64 || max_batch.load() > 1);  // requests were actually combined
    std::printf("flat combining: %d threads x %d rounds -> %ld batches (largest %ld)\n", nthreads, rounds, batches.load(),
                max_batch.load());
}

// ---- 2. buffer pool: concurrent create / destroy of "models", cap, trim, allocation failures --------------------------
static void test_pool_concurrent(size_t cap_mb, int fail_every, int nthreads, int iters)
{
    stub::fail_every = fail_every;
    const long a0 = stub::allocs, f0 = stub::frees;
    {
        BigPool pool(cap_mb << 20);
        std::atomic<long> ooms{0};
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([&, t] {
                stub::set_device(t % 3);  // three "devices": a parked buffer only goes back to its own device
                for (int i = 0; i < iters; ++i) {
                    // a model: kernel matrix, inverse factor (pooled sizes), a small vector block (not pooled)
                    const size_t big = ((size_t)64 << 20) * (1 + (size_t)((t + i) % 4));
                    void *K = nullptr, *X = nullptr, *v = nullptr;
                    const int r1 = pool.alloc(&K, big), r2 = pool.alloc(&X, big), r3 = pool.alloc(&v, 4096);
                    if (r1 || r2 || r3)
                        ++ooms;  // create() fails with GPX_E_OOM and releases what it got -- nothing may leak
                    CHECK((r1 == 0) == (K != nullptr) && (r2 == 0) == (X != nullptr) && (r3 == 0) == (v != nullptr));
                    if (i % 7 == 3)
                        pool.trim();  // gpx_trim() from another thread's point of view
                    pool.release(K);
                    pool.release(X);
                    pool.release(v);
                    CHECK(pool.parked_bytes() <= (cap_mb << 20));
                }
            });
        for (auto &t : th)
            t.join();
        CHECK(pool.live_buffers() == 0);
        CHECK(pool.parked_bytes() <= (cap_mb << 20));
        if (cap_mb == 0)
            CHECK(pool.parked_bytes() == 0);
        if (fail_every == 0)
            CHECK(ooms.load() == 0);
        pool.trim();
        CHECK(pool.parked_bytes() == 0);
        std::printf("pool: cap %zu MiB, fail every %d: %ld allocations, %ld failed creates\n", cap_mb, fail_every,
                    stub::allocs - a0, ooms.load());
    }
    CHECK(stub::live_bytes.load() == 0);  // every device buffer went back
    CHECK((stub::allocs - a0) >= (stub::frees - f0));
    stub::fail_every = 0;
}

static void test_pool_oom_paths()
{
    // GPX_POOL_MB=0 (nothing is ever parked) + an allocator that always fails: alloc reports out of memory, hands back
    // a null pointer, leaves nothing registered, and a later success works
    BigPool pool(0);
    stub::fail_always_oom = 1;
    void *p = (void *)0x1;
    CHECK(pool.alloc(&p, (size_t)1 << 30) == 1 && p == nullptr);
    CHECK(pool.alloc(&p, 128) == 1 && p == nullptr);
    CHECK(pool.live_buffers() == 0 && pool.parked_bytes() == 0);
    stub::fail_always_oom = 0;
    CHECK(pool.alloc(&p, (size_t)1 << 30) == 0 && p);
    pool.release(p);
    CHECK(pool.parked_bytes() == 0 && stub::live_bytes.load() == 0);
    // out of memory WITH buffers parked: the pool empties itself and the retry succeeds
    BigPool pool2((size_t)8 << 30);
    void *a = nullptr, *b = nullptr;
    CHECK(pool2.alloc(&a, (size_t)1 << 30) == 0);
    pool2.release(a);
    CHECK(pool2.parked_bytes() == ((size_t)1 << 30));
    const long k = stub::allocs.load() + 1;
    stub::fail_every = (int)k;  // allocation number k (the next one) fails, k + 1 does not
    CHECK(pool2.alloc(&b, (size_t)3 << 30) == 0 && b);  // first try fails, the pool trims itself, the retry succeeds
    stub::fail_every = 0;
    CHECK(pool2.parked_bytes() == 0);
    pool2.release(b);
    pool2.trim();
    CHECK(stub::live_bytes.load() == 0);
    std::printf("pool: out-of-memory paths ok\n");
}

// ---- 3. per-device once flags ---------------------------------------------------------------------------------------------
static void test_per_device_once(int nthreads)
{
    PerDeviceOnce once;
    std::atomic<int> count[4] = {{0}, {0}, {0}, {0}};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            stub::set_device(t % 4);
            for (int i = 0; i < 50; ++i)
                once.run([&] { ++count[t % 4]; });
        });
    for (auto &t : th)
        t.join();
    for (int d = 0; d < 4; ++d)
        CHECK(count[d].load() == 1);
    std::printf("per-device once: 4 devices, %d threads ok\n", nthreads);
}

// ---- 4. Eigen's pivot order ------------------------------------------------------------------------------------------------
static void test_pivot_order()
{
    std::vector<int> perm;
    eigen_pivot_order({1.1, 1.1, 1.1, 1.1}, perm);  // uniform diagonal: identity
    CHECK((perm == std::vector<int>{0, 1, 2, 3}));
    eigen_pivot_order({1.0, 3.0, 2.0, 3.0, -5.0}, perm);  // largest |d| first; ties: the FIRST; swaps, not a sort
    // step 0: |-5| at 4 <-> 0: d = [-5, 3, 2, 3, 1], perm = [4, 1, 2, 3, 0]; step 1: 3 at 1 stays; step 2: 3 at 3 <-> 2
    CHECK((perm == std::vector<int>{4, 1, 3, 2, 0}));
    eigen_pivot_order({}, perm);
    CHECK(perm.empty());
    std::vector<double> big(5000);
    for (size_t i = 0; i < big.size(); ++i)
        big[i] = 1.0 + (double)((i * 7919) % 101) * 0.01;
    eigen_pivot_order(big, perm);
    std::vector<int> sorted(perm);
    std::sort(sorted.begin(), sorted.end());
    for (size_t i = 0; i < sorted.size(); ++i)
        CHECK(sorted[i] == (int)i);  // a permutation
    for (size_t i = 1; i < perm.size(); ++i)
        CHECK(big[perm[i - 1]] >= big[perm[i]]);  // non-increasing pivots
    std::printf("pivot order ok\n");
}

int main(int argc, char **argv)
{
    const int nthreads = argc > 1 ? std::atoi(argv[1]) : 841;
    set_device_backend(&stub::backend);
    test_pivot_order();
    test_per_device_once(64);
    test_flat_combining(nthreads, 3);
    test_pool_concurrent(16384, 0, 16, 200);  // the default cap
    test_pool_concurrent(256, 0, 16, 200);    // a cap that is hit all the time
    test_pool_concurrent(0, 0, 8, 100);       // GPX_POOL_MB=0: the pool is off
    test_pool_concurrent(1024, 5, 16, 200);   // every fifth device allocation fails
    test_pool_oom_paths();
    CHECK(stub::live_bytes.load() == 0);
    std::printf("host concurrency ok\n");
    return 0;
}
