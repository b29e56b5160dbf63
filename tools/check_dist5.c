// tools/check_dist5.c -- finds the smallest double s with (float)sqrt(s) >= 5.0f, i.e. the exact squared-distance
// threshold URF_DIST5_SQ used by k_ring instead of "d = (float)sqrt(s); d < 5.0".   gcc -O2 tools/check_dist5.c -lm
#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
static int pred(double s){ float d=(float)sqrt(s); return (double)d < 5.0; }
int main(){
  double lo=24.0, hi=26.0; // pred(lo)=1, pred(hi)=0
  uint64_t a,b; memcpy(&a,&lo,8); memcpy(&b,&hi,8);
  while(b-a>1){ uint64_t m=a+(b-a)/2; double x; memcpy(&x,&m,8); if(pred(x)) a=m; else b=m; }
  double S; memcpy(&S,&b,8);
  printf("S* = %a = %.20g  pred(S*)=%d pred(prev)=%d\n", S, S, pred(S), pred(nextafter(S,0)));
  // monotonic sanity around
  long bad=0; double x=nextafter(S,0); for(int i=0;i<1000000;i++){ if(!pred(x)) bad++; x=nextafter(x,0);} x=S; for(int i=0;i<1000000;i++){ if(pred(x)) bad++; x=nextafter(x,1e9);} printf("bad=%ld\n",bad);
  return 0;}
