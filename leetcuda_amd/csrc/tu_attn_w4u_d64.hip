// tu_attn_w4u_d64.hip — translation unit of the merged-phase attention kernel (attn_w4u.hip), D = 64, V as [B,H,N,D] — see lc_launch.h
#define W4U_D 64
#define W4U_VT false
#define W4U_TAG d64
#include "tu_attn_w4u_impl.h"
