// hpt_rccl_check.cpp — compiles to nothing.  It exists so that the build FAILS when the hand-written RCCL declarations of hpt_rccl_abi.h (what
// hpt_multi.hip calls through dlsym'd pointers) stop matching the <rccl/rccl.h> of the ROCm the library is built against: constants,
// the by-value 128-byte id, enum sizes, and every entry point's parameter list.
#include <rccl/rccl.h>
#include <type_traits>
#include "hpt_rccl_abi.h"

static_assert(HPT_NCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES && sizeof(hpt_nccl_unique_id) == sizeof(ncclUniqueId) && alignof(hpt_nccl_unique_id) == alignof(ncclUniqueId), "ncclUniqueId");
static_assert(HPT_NCCL_SUCCESS == (int)ncclSuccess, "ncclSuccess");
static_assert(HPT_NCCL_FLOAT32 == (int)ncclFloat32, "ncclFloat32");
static_assert(HPT_NCCL_SUM == (int)ncclSum, "ncclSum");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int), "RCCL enums travel as int");
static_assert(sizeof(ncclComm_t) == sizeof(hpt_nccl_comm_t), "ncclComm_t is a pointer");
// the parameter lists, with RCCL's own types (an added, removed or reordered parameter fails here)
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId *)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommInitAll), ncclResult_t (*)(ncclComm_t *, int, const int *)>::value, "ncclCommInitAll");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclCommCount), ncclResult_t (*)(const ncclComm_t, int *)>::value, "ncclCommCount");
static_assert(std::is_same<decltype(&ncclGroupStart), ncclResult_t (*)()>::value && std::is_same<decltype(&ncclGroupEnd), ncclResult_t (*)()>::value, "ncclGroupStart / End");
static_assert(std::is_same<decltype(&ncclSend), ncclResult_t (*)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclSend");
static_assert(std::is_same<decltype(&ncclRecv), ncclResult_t (*)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclRecv");
static_assert(std::is_same<decltype(&ncclReduce), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t)>::value, "ncclReduce");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char *(*)(ncclResult_t)>::value, "ncclGetErrorString");
