/* htslib/khash.h — SHIM: a string-keyed map with the khash macro surface that bamreadcount.cpp:42-44,559-580 uses
 * (KHASH_MAP_INIT_STR, kh_init, kh_put, kh_get, kh_value, kh_end).  Own code (std::map inside), not klib. */
#ifndef BRC_REF_SHIM_KHASH_H
#define BRC_REF_SHIM_KHASH_H
#include <map>
#include <string>
#include <vector>
typedef unsigned khiter_t;
typedef unsigned khint_t;
template <class V>
struct brc_shim_khash {
    std::map<std::string, unsigned> index;
    std::vector<V> vals;
};
#define KHASH_MAP_INIT_STR(name, khval_t)                                                                       \
    typedef khval_t kh_##name##_val_t;                                                                           \
    typedef brc_shim_khash<kh_##name##_val_t> kh_##name##_t;                                                               \
    static inline kh_##name##_t* kh_init_##name() { return new kh_##name##_t(); }                                \
    static inline void kh_destroy_##name(kh_##name##_t* h) { delete h; }                                         \
    static inline khiter_t kh_put_##name(kh_##name##_t* h, const char* key, int* ret) {                          \
        std::map<std::string, unsigned>::iterator it = h->index.find(key);                                       \
        if (it != h->index.end()) { *ret = 0; return it->second; }                                               \
        h->index[key] = (unsigned)h->vals.size(); h->vals.push_back(kh_##name##_val_t()); *ret = 1;                        \
        return (khiter_t)(h->vals.size() - 1);                                                                   \
    }                                                                                                            \
    static inline khiter_t kh_get_##name(const kh_##name##_t* h, const char* key) {                              \
        std::map<std::string, unsigned>::const_iterator it = h->index.find(key);                                 \
        return it == h->index.end() ? (khiter_t)h->vals.size() : it->second;                                     \
    }
#define khash_t(name) kh_##name##_t
#define kh_init(name) kh_init_##name()
#define kh_destroy(name, h) kh_destroy_##name(h)
#define kh_put(name, h, k, r) kh_put_##name(h, k, r)
#define kh_get(name, h, k) kh_get_##name(h, k)
#define kh_value(h, x) ((h)->vals[x])
#define kh_end(h) ((khiter_t)(h)->vals.size())
#endif
