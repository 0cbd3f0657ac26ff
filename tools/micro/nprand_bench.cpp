// times ssw_np_permutation_prefix(1 560 000, 10 000) through the built library: g++ -O2 nprand_bench.cpp -o nprand_bench -ldl && ./nprand_bench ../../seesaw_amd/libseesaw_hip.so
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <vector>
#include <dlfcn.h>
typedef int (*fn_t)(uint32_t*, int32_t*, int64_t, int64_t, int64_t*);
int main(int argc,char**argv){
  void*h=dlopen(argv[1],RTLD_NOW); if(!h){printf("%s\n",dlerror());return 1;}
  fn_t f=(fn_t)dlsym(h,"ssw_np_permutation_prefix");
  std::vector<uint32_t> key(624); for(int i=0;i<624;i++) key[i]=i*2654435761u+12345;
  int32_t pos=624; std::vector<int64_t> out(10000);
  double best=1e9; 
  for(int r=0;r<15;r++){auto t=std::chrono::steady_clock::now(); f(key.data(),&pos,1560000,10000,out.data());
    double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t).count(); if(ms<best)best=ms;}
  printf("best %.3f ms  out0 %ld out9999 %ld pos %d\n",best,(long)out[0],(long)out[9999],pos);
}
