// raw_vector.h — std::vector whose resize() leaves new elements uninitialised (every element is overwritten by its producer).
#pragma once
#include <atomic>
#include <memory>
#include <utility>
#include <vector>

// Called with every block a raw_vector gives back, before it is freed: the program page-locks the columns it decodes into
// (driver.cpp) and must release the lock of a block that a growing vector is about to free.
// (atomic: set by the program's main thread while reader and worker threads give blocks back)
inline std::atomic<void (*)(void*)> bqc_raw_vector_free_hook{nullptr};
// The other direction: a producer that fills a buffer by copies from the device (the BGZF reader when the GPU inflates) asks for
// the block to be page-locked; whoever sets this hook sets the one above too.
inline std::atomic<void (*)(const void*, size_t)> bqc_raw_vector_pin_hook{nullptr};

template <typename T>
struct no_init_alloc : std::allocator<T> {
    template <typename U> struct rebind { using other = no_init_alloc<U>; };
    void deallocate(T* p, std::size_t n)
    {
        if (auto hook = bqc_raw_vector_free_hook.load(std::memory_order_acquire)) hook((void*)p);
        std::allocator<T>::deallocate(p, n);
    }
    template <typename U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <typename U, typename... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <typename T> using raw_vector = std::vector<T, no_init_alloc<T>>;

// Large streaming buffers on transparent huge pages where the kernel offers them on request (THP mode "madvise"): fewer page
// faults when a buffer is first filled and fewer TLB misses while gigabytes stream through it.  Call after the vector got its
// capacity and before the memory is touched; a no-op for small buffers and elsewhere.
#include <sys/mman.h>
#include <cstdint>
#include <cstdlib>
template <typename V>
inline void advise_huge(V& v)
{
#if defined(MADV_HUGEPAGE)
    static const bool off = getenv("BQC_NO_HUGEPAGES") != nullptr; // (on a badly fragmented host a huge-page fault may stall in compaction)
    if (off) return;
    const size_t bytes = v.capacity() * sizeof(typename V::value_type);
    if (bytes < (8u << 20)) return;
    const uintptr_t a = ((uintptr_t)v.data() + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), z = ((uintptr_t)v.data() + bytes) & ~(uintptr_t)((2u << 20) - 1);
    if (z > a) (void)madvise((void*)a, (size_t)(z - a), MADV_HUGEPAGE);
#else
    (void)v;
#endif
}
