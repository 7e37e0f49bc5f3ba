// hpt_rccl_abi.h — the few RCCL declarations hpt_multi.hip binds at run time (dlopen("librccl.so.1"): the library has no link-time
// dependency on RCCL and a single-GPU host never loads it), written out by hand — and CHECKED at build time against <rccl/rccl.h> by
// hpt_rccl_check.cpp ("values look right, but nothing has ever checked them against the header").
#ifndef HPT_RCCL_ABI_H
#define HPT_RCCL_ABI_H
#include <hip/hip_runtime_api.h>
#include <cstddef>

#define HPT_NCCL_UNIQUE_ID_BYTES 128
#define HPT_NCCL_SUCCESS 0      /* ncclResult_t  ncclSuccess */
#define HPT_NCCL_FLOAT32 7      /* ncclDataType_t ncclFloat32 */
#define HPT_NCCL_SUM 0          /* ncclRedOp_t   ncclSum */

struct hpt_nccl_comm;                                           // ncclComm (opaque)
typedef struct hpt_nccl_comm *hpt_nccl_comm_t;                  // ncclComm_t
typedef struct { char internal[HPT_NCCL_UNIQUE_ID_BYTES]; } hpt_nccl_unique_id;   // ncclUniqueId, passed BY VALUE to ncclCommInitRank
// (the enum parameters — ncclDataType_t, ncclRedOp_t — and the ncclResult_t return value travel as int: the check asserts their size)
typedef int (*hpt_nccl_get_unique_id_fn)(hpt_nccl_unique_id *);
typedef int (*hpt_nccl_comm_init_rank_fn)(hpt_nccl_comm_t *, int, hpt_nccl_unique_id, int);
typedef int (*hpt_nccl_comm_init_all_fn)(hpt_nccl_comm_t *, int, const int *);
typedef int (*hpt_nccl_comm_destroy_fn)(hpt_nccl_comm_t);
typedef int (*hpt_nccl_comm_count_fn)(const hpt_nccl_comm_t, int *);
typedef int (*hpt_nccl_group_fn)();
typedef int (*hpt_nccl_send_fn)(const void *, size_t, int, int, hpt_nccl_comm_t, hipStream_t);
typedef int (*hpt_nccl_recv_fn)(void *, size_t, int, int, hpt_nccl_comm_t, hipStream_t);
typedef int (*hpt_nccl_reduce_fn)(const void *, void *, size_t, int, int, int, hpt_nccl_comm_t, hipStream_t);
typedef const char *(*hpt_nccl_get_error_string_fn)(int);
#endif
