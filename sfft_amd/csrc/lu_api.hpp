// lu_api.hpp -- host-side interface of the pivoted-LU kernels (lu.hpp), which live in a translation unit of their own (lu.hip: the
// panel kernel has one instantiation per tile shape and takes minutes to compile).  Included by sfft_amd.hip and lu.hip.
#ifndef SFFT_AMD_LU_API_HPP
#define SFFT_AMD_LU_API_HPP
#include <hip/hip_runtime.h>

#define LU_NB 64
#define LU_NT 256
#define LU_MAXTOUCH (2 * LU_NB)
#define LU_MAX_ROWS (LU_NT * 6 * 16)  // rows of the largest panel: 16 workgroups x 256 threads x 6 rows

struct LuPerm { int count; int pad[3]; int pos[LU_MAXTOUCH]; int src[LU_MAXTOUCH]; };     // one per panel: row pos receives row src

// The whole factorisation P A = L U of the bordered system (column n = right-hand side) as a chain of launches on `s`; `perms` holds
// LU_PERMS_PER_PANEL lists per 64-column panel.  note(name) is called once per launch with the kernel's name (may be null).
#define LU_PERMS_PER_PANEL 1
typedef void (*lu_note_fn)(const char* kernel_name);
// xchg: lu_xchg_bytes() bytes of zero-initialised device memory (the hand-off slots of the multi-workgroup panel; needed for n > 1024);
// epoch_ctr: the device counter chol_begin advances once per solve.
size_t lu_xchg_bytes();
void lu_factor_launches(double* A, int ld, int n, LuPerm* perms, int* status, double* rd, void* xchg, const unsigned int* epoch_ctr, hipStream_t s, lu_note_fn note);
#endif
