// tu_attn_w4u_d128.hip — translation unit of the merged-phase attention kernel (attn_w4u.hip), D = 128, V as [B,H,N,D] — see lc_launch.h
#define W4U_D 128
#define W4U_VT false
#define W4U_TAG d128
#include "tu_attn_w4u_impl.h"
