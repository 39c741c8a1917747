#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const float* a, const float* b, float* o, int n) {
  int i = blockIdx.x*blockDim.x+threadIdx.x; if (i>=n) return;
  float x=a[i], y=b[i];
  o[i*8+0] = x/y;
  o[i*8+1] = (float)sqrt((double)fabsf(x));
  o[i*8+2] = x/3.0f;
  o[i*8+3] = x/6.0f;
  o[i*8+4] = x*y + x;      // contraction check
  o[i*8+5] = (float)acos((double)fminf(fmaxf(x,-1.f),1.f));
  o[i*8+6] = (float)cos((double)x);
  o[i*8+7] = sqrtf(fabsf(x));
}
int main(){
  int n=1<<20; std::vector<float> a(n),b(n),o(n*8);
  std::mt19937 g(1); std::normal_distribution<float> d(0,1);
  for(int i=0;i<n;i++){a[i]=d(g)*powf(10.f,(i%13)-6); b[i]=d(g)*powf(10.f,((i/13)%13)-6); if(b[i]==0)b[i]=1;}
  float *da,*db,*dd; hipMalloc(&da,n*4);hipMalloc(&db,n*4);hipMalloc(&dd,n*32);
  hipMemcpy(da,a.data(),n*4,hipMemcpyHostToDevice);hipMemcpy(db,b.data(),n*4,hipMemcpyHostToDevice);
  k<<<n/256,256>>>(da,db,dd,n); hipMemcpy(o.data(),dd,n*32,hipMemcpyDeviceToHost);
  long bad[8]={0};
  for(int i=0;i<n;i++){ float x=a[i],y=b[i];
    float r[8]; r[0]=x/y; r[1]=(float)sqrt((double)fabsf(x)); r[2]=x/3.0f; r[3]=x/6.0f; volatile float t=x*y; r[4]=t+x;
    r[5]=(float)acos((double)fminf(fmaxf(x,-1.f),1.f)); r[6]=(float)cos((double)x); r[7]=sqrtf(fabsf(x));
    for(int k2=0;k2<8;k2++) if(!(r[k2]==o[i*8+k2]) && !(std::isnan(r[k2])&&std::isnan(o[i*8+k2]))) { if(bad[k2]<3) printf("k=%d x=%a y=%a cpu=%a gpu=%a\n",k2,x,y,r[k2],o[i*8+k2]); bad[k2]++; }
  }
  for(int k2=0;k2<8;k2++) printf("op %d mismatches %ld\n",k2,bad[k2]);
}
