// CPU model of icc_pow_pos (csrc/write_kernels.hip): the same table, polynomial and splitting, with glibc exp2f standing in for
// v_exp_f32, against libm pow() in double on 4 M points per exponent.  gcc -O2 tools/fastpow_check.c -lm && ./a.out
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static float Tc[128]; static double TL[128];
static void init(void){ for(int i=0;i<128;i++){ double center = 0.5 + (i + 0.5)/256.0; float c = (float)(1.0/center); Tc[i]=c; TL[i] = -log2((double)c);} }
static inline uint32_t fbits(float f){uint32_t u; memcpy(&u,&f,4); return u;}
// x > 0 double (may carry more than 24 bits), y > 0
static float fast_pow(double x, double y){
  float xf = (float)x;
  int e; float m = frexpf(xf,&e);          // [0.5,1)
  int idx = (fbits(m)>>16)&0x7f;
  float c = Tc[idx];
  float r = fmaf(m,c,-1.0f);
  float dx = (float)(x - (double)xf);      // residual of the float conversion
  r = fmaf(ldexpf(dx,-e), c, r);
  // log2(1+r) = r*(k1 + r*(k2 + r*(k3 + r*(k4 + r*k5))))
  const float k1=1.4426950408889634f,k2=-0.7213475204444817f,k3=0.4808983469629878f,k4=-0.36067376022224085f,k5=0.2885390081777927f;
  float p = r*fmaf(r,fmaf(r,fmaf(r,fmaf(r,k5,k4),k3),k2),k1);
  double t = y*((double)e + TL[idx] + (double)p);
  double n = rint(t);
  float f = (float)(t-n);
  return ldexpf(exp2f(f),(int)n);
}
int main(){ init(); srand(1);
  double ys[]={2.4,1.0/2.4,2.2,1.8,1.0/2.2,2.6,0.45};
  for(int k=0;k<7;k++){ double y=ys[k]; long bad=0,n=0; double maxrel=0; 
    for(long i=0;i<4000000;i++){ double u=(rand()+0.5)/RAND_MAX; double x; 
      if(i&1){ x=(double)(float)u; } else { x = u*0.9478672985781991 + 0.0521327014218009; if(i%4==0) x=exp(-20*u); }
      float want=(float)pow(x,y); float got=fast_pow(x,y); 
      double rel=fabs((double)got-pow(x,y))/pow(x,y); if(rel>maxrel)maxrel=rel; if(want!=got)bad++; n++; }
    printf("y=%.4f float-mismatch %.3f%% max rel err %.3e (float ulp 5.96e-8)\n",y,100.0*bad/n,maxrel);}
  return 0;}
