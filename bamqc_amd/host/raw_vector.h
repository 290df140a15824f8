// raw_vector.h — std::vector whose resize() leaves new elements uninitialised (every element is overwritten by its producer).
#pragma once
#include <memory>
#include <utility>
#include <vector>

template <typename T>
struct no_init_alloc : std::allocator<T> {
    template <typename U> struct rebind { using other = no_init_alloc<U>; };
    template <typename U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <typename U, typename... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <typename T> using raw_vector = std::vector<T, no_init_alloc<T>>;
