// tools/check_div_pi.c -- exhaustive host check that the 3-operation division by pi used on the device
// (urf_div_pi, urban_road_filter_amd/csrc/urf_device.hpp) equals the IEEE division for every float in [0, 600].
// gcc -O2 -fopenmp -mfma -ffp-contract=off tools/check_div_pi.c -o /tmp/check_div_pi -lm && /tmp/check_div_pi
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <omp.h>
int main(){
  const double PI=0x1.921fb54442d18p+1; const double RPI=1.0/PI; // RN(1/pi)
  float top=600.0f; uint32_t tb; memcpy(&tb,&top,4);
  long bad=0;
  #pragma omp parallel for reduction(+:bad) schedule(static)
  for(uint32_t b=0;b<=tb;b++){ float f; memcpy(&f,&b,4); double a=(double)f;
    double q=a*RPI; double r=__builtin_fma(-q,PI,a); double q2=__builtin_fma(r,RPI,q);
    if(q2!=a/PI) bad++; }
  printf("1/pi=%a checked %u floats, mismatches %ld\n",RPI,tb+1,bad);
  // also 180-t style not needed
  return 0;}
