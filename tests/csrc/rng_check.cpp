#include <cstdio>
#include <cmath>
#include "rng_chacha.h"
int main(){
  // RFC 8439 2.3.2
  uint32_t key[8]; for(int i=0;i<8;i++) key[i]= (4*i) | ((4*i+1)<<8) | ((4*i+2)<<16) | ((uint32_t)(4*i+3)<<24);
  uint32_t nonce[3]={0x09000000,0x4a000000,0};
  uint32_t out[16]; mktrng::chacha20_block(key,1,nonce,out);
  for(int i=0;i<16;i++) printf("%08x ",out[i]); printf("\n");
  mktrng::Rng r(key,0,1);
  double s=0,s2=0,s4=0; int n=4000000; double mx=0; int tail5=0;
  for(int i=0;i<n;i++){ double g=r.gauss(); s+=g; s2+=g*g; s4+=g*g*g*g; if(fabs(g)>mx) mx=fabs(g); if (fabs(g)>4) tail5++; }
  printf("mean %g var %g kurt %g max %g P(|g|>4) %g (exp 6.33e-5)\n", s/n, s2/n, s4/n/(s2/n)/(s2/n), mx, (double)tail5/n);
  // accuracy vs libm
  double worst=0; mktrng::Rng q(key,0,2);
  for(int i=0;i<200000;i++){ uint64_t a=q.next(), b=q.next(); double u1=((a>>11)+1)*0x1p-53,u2=(b>>11)*0x1p-53; double ref=sqrt(-2*log(u1))*cos(2*M_PI*u2); double e=fabs(ref-mktrng::box_muller(a,b)); if(e>worst) worst=e; }
  printf("worst abs err vs libm %g\n", worst);
}
