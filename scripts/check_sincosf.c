#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static inline uint32_t asuint(float f){uint32_t u; memcpy(&u,&f,4); return u;}
static inline uint32_t abstop12(float x){return (asuint(x)>>20)&0x7ff;}
static inline void poly(double x, double x2, const sincos_t *p, int n, float *sinp, float *cosp){
  double x3,x4,x5,x6,s,c,c1,c2,s1;
  x4=x2*x2; x3=x2*x; c2=p->c3+x2*p->c4; s1=p->s2+x2*p->s3;
  float *tmp=(n&1?cosp:sinp); cosp=(n&1?sinp:cosp); sinp=tmp;
  c1=p->c0+x2*p->c1; x5=x3*x2; x6=x4*x2; s=x+x3*p->s1; c=c1+x4*p->c2;
  *sinp=s+x5*s1; *cosp=c+x6*c2;
}
static inline double reduce_fast(double x, const sincos_t *p, int *np){
  double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return x-n*p->hpi;
}
void my_sincosf(float y, float *sinp, float *cosp){
  double x=y, s; int n; const sincos_t *p=&T[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    s=x*x;
    if (abstop12(y) < abstop12(0x1p-12f)) { *sinp=y; *cosp=1.0f; return; }
    poly(x,s,p,0,sinp,cosp);
  } else {
    x=reduce_fast(x,p,&n); s=p->sign[n&3]; if (n&2) p=&T[1];
    poly(x*s,x*x,p,n,sinp,cosp);
  }
}
int main(){
  uint64_t bad=0,tot=0; float hi=8.0f;
  for (uint32_t u=0; u<=asuint(hi); ++u){
    for (int sg=0; sg<2; ++sg){ uint32_t v=u|(sg?0x80000000u:0); float x; memcpy(&x,&v,4);
      float s0,c0,s1,c1; sincosf(x,&s0,&c0); my_sincosf(x,&s1,&c1); tot++;
      if (asuint(s0)!=asuint(s1)||asuint(c0)!=asuint(c1)){ if(bad<10) printf("x=%a glibc %a %a mine %a %a\n",x,s0,c0,s1,c1); bad++; }
      float s2=sinf(x), c2=cosf(x); if (asuint(s2)!=asuint(s0)||asuint(c2)!=asuint(c0)) { static int w=0; if(w++<5) printf("sinf/cosf != sincosf at %a\n",x);} }
  }
  printf("total %llu mismatches %llu\n",(unsigned long long)tot,(unsigned long long)bad); return 0; }
