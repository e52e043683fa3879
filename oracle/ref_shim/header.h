/* header.h — SHIM of htslib-1.10's header.h (sam_hrecs_t and friends), only what find_library_names
 * (bamreadcount.cpp:92-111) touches.  Test infrastructure; see sam.h in this directory. */
#ifndef BRC_REF_SHIM_HEADER_H
#define BRC_REF_SHIM_HEADER_H
#include "sam.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct sam_hrec_tag_s {
    struct sam_hrec_tag_s* next;
    const char* str; /* "XX:value" */
    int len;
} sam_hrec_tag_t;
typedef struct sam_hrec_type_s {
    struct sam_hrec_type_s *next, *prev, *global_next, *global_prev;
    sam_hrec_tag_t* tag;
    int type;
} sam_hrec_type_t;
typedef struct {
    char* name;
    sam_hrec_type_t* ty;
    int name_len, id;
} sam_hrec_rg_t;
typedef struct sam_hrecs_t {
    int nrg;
    sam_hrec_rg_t* rg;
} sam_hrecs_t;
typedef struct sam_hdr_shim_t {
    sam_hrecs_t* hrecs;
} sam_hdr_t;
sam_hdr_t* sam_hdr_parse(size_t l_text, const char* text);
#ifdef __cplusplus
}
#endif
#endif
