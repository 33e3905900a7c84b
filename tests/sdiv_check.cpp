// TEST INFRASTRUCTURE.  wm_udiv (rtl-wmbus_amd/csrc/wm_k2_rla.h) divides through the hardware's approximate reciprocal (within 1 ulp)
// and one +-1 repair.  The same formula here, with the reciprocal deliberately pushed 1 ulp either way, over every divisor of the
// domain and every dividend next to a multiple of it (where a wrong quotient would show): the repair always lands on a / b.
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
static float nudge(float x, int d){ uint32_t u; memcpy(&u,&x,4); u+=d; memcpy(&x,&u,4); return x; }
int main(){
  uint64_t bad=0, n=0;
  for (unsigned b=4;b<4096;b++) for (int d=-1; d<=1; d++) {
    float rc = nudge(1.0f/(float)b, d);
    for (unsigned q=0; (uint64_t)q*b < (1u<<24); q++) for (int off=-1; off<=1; off++) {
      long long a=(long long)q*b+off; if (a<0||a>=(1<<24)) continue;
      unsigned ua=(unsigned)a; unsigned qq=(unsigned)((float)ua*rc); int r=(int)ua-(int)(qq*b);
      qq = r<0? qq-1u : r>=(int)b ? qq+1u : qq; n++;
      if (qq != ua/b) bad++;
    }
  }
  printf("%llu cases, %llu wrong\n",(unsigned long long)n,(unsigned long long)bad); return bad!=0;
}
