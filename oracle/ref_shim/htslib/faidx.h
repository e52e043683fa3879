/* htslib/faidx.h — SHIM (see ../sam.h) */
#ifndef BRC_REF_SHIM_FAIDX_H
#define BRC_REF_SHIM_FAIDX_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct faidx_shim_t faidx_t;
faidx_t* fai_load(const char* fn);
void fai_destroy(faidx_t* fai);
/* whole sequence `reg` (a contig name), malloc()ed and NUL-terminated; *len = its length */
char* fai_fetch(const faidx_t* fai, const char* reg, int* len);
#ifdef __cplusplus
}
#endif
#endif
