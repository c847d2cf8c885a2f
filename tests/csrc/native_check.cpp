// exhaustive-ish host check: the magic-number conversion equals the reference form for every class of input
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
static uint32_t lowd(double v){ double s = v + 4503599627370496.0; uint64_t b; memcpy(&b,&s,8); return (uint32_t)b; }
static uint32_t ref32(double x){ x -= floor(x*2.3283064365386963e-10)*4.294967296e9; return x==4.294967296e9?0u:(uint32_t)x; }
static uint32_t new32(double x){ x -= floor(x*2.3283064365386963e-10)*4.294967296e9; return lowd(trunc(x)); }
static uint64_t ref64(double x){ x -= floor(x*5.421010862427522e-20)*1.8446744073709552e19; return x==1.8446744073709552e19?0ull:(uint64_t)x; }
static uint64_t new64(double x){ x -= floor(x*5.421010862427522e-20)*1.8446744073709552e19; double hi=trunc(x*2.3283064365386963e-10); double lo=trunc(x-hi*4.294967296e9); return ((uint64_t)lowd(hi)<<32)|lowd(lo); }
int main(){
  std::mt19937_64 g(7); long bad=0, n=0;
  auto test=[&](double x){ n++; if(ref32(x)!=new32(x)){ if(bad<5) printf("u32 mismatch x=%a ref %u new %u\n",x,ref32(x),new32(x)); bad++;} if(ref64(x)!=new64(x)){ if(bad<5) printf("u64 mismatch x=%a ref %llu new %llu\n",x,(unsigned long long)ref64(x),(unsigned long long)new64(x)); bad++;} };
  double specials[]={0.0,-0.0,0.3,-0.3,-1e-300,1e-300,-0.5,0.5,1.0,-1.0,4294967295.0,4294967296.0,4294967297.0,-4294967296.0,4294967295.7,-4294967295.7,
    1.8446744073709552e19,-1.8446744073709552e19,1.8446744073709550e19,9.2233720368547758e18,-9.2233720368547758e18,1.8446744073709552e19*3,-1e-20, -1024.0,-1023.9999, 4.5e15, -4.5e15};
  for(double x:specials){ test(x); test(nextafter(x,1e300)); test(nextafter(x,-1e300)); }
  for(int e=-60;e<120;e++) for(int i=0;i<200000;i++){ uint64_t m=g(); double f=(double)(m>>11)*0x1p-53+0.5; double x=ldexp(f,e); if(m&1) x=-x; test(x); if((i&7)==0) test(floor(x)); }
  // values just below multiples of 2^W (negative tiny offsets) that round up to 2^W
  for(int i=0;i<2000000;i++){ double k=(double)(int)(g()%2001-1000); double eps=ldexp((double)(g()>>11)*0x1p-53, -(int)(g()%80)); test(k*4.294967296e9-eps); test(k*1.8446744073709552e19-eps*4e9); test(k*4.294967296e9+eps); }
  printf("%ld cases, %ld mismatches\n", n, bad); return bad!=0; }
