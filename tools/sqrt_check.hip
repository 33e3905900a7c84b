// exhaustive check: device sqrt/div used by wm_exact.h vs host IEEE results on the RSSI operand domain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../rtl-wmbus_amd/csrc/wm_exact.h"
__global__ void k(const float* x, float* y, int n){int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i]= (blockIdx.y==0)? __builtin_sqrtf(x[i]) : wm_sqrt(x[i]);}
int main(){
  std::vector<float> x; for(int sc: {64,256}) for(long v=0; v<= (sc==64? 2*1016L*1016L : 2*2880L*2880L); v+= (sc==64?1:7)) x.push_back((float)v/sc);
  int n=x.size(); float *dx,*dy; hipMalloc(&dx,n*4); hipMalloc(&dy,n*4); hipMemcpy(dx,x.data(),n*4,hipMemcpyHostToDevice);
  k<<<(n+255)/256,256>>>(dx,dy,n); std::vector<float> y(n); hipMemcpy(y.data(),dy,n*4,hipMemcpyDeviceToHost);
  long bad=0; for(int i=0;i<n;i++){ float r=sqrtf(x[i]); if(memcmp(&r,&y[i],4)){ if(bad<5) printf("x=%a host=%a dev=%a\n",x[i],r,y[i]); bad++; } }
  printf("sqrt: %d values, %ld mismatches\n",n,bad); return 0; }
