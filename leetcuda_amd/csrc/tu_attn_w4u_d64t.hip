// tu_attn_w4u_d64t.hip — translation unit of the merged-phase attention kernel (attn_w4u.hip), D = 64, V as [B,H,D,N] — see lc_launch.h
#define W4U_D 64
#define W4U_VT true
#define W4U_TAG d64t
#include "tu_attn_w4u_impl.h"
