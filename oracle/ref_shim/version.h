/* version.h — stands in for the reference's generated version header (build-common/cmake) */
#ifndef BRC_REF_SHIM_VERSION_H
#define BRC_REF_SHIM_VERSION_H
static const char __g_prog_version[] = "1.0.1-refshim";
static const char __g_commit_hash[] = "reference sources compiled against oracle/ref_shim";
#endif
