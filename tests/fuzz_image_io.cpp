// Mutation fuzzer for the native image decoders / PNG encoder (CPU only, test infrastructure).
// Built with -fsanitize=address,undefined by tests/test_image_fuzz.py: decoders must reject or
// decode every mutated file without touching memory out of bounds.
//   usage: fuzz_image_io <iterations per file> <file>...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include "image_io.hpp"
using namespace hmrm;
static std::vector<uint8_t> readf(const char* p){ FILE* f=fopen(p,"rb"); std::vector<uint8_t> b; if(!f) return b; uint8_t c[65536]; size_t g; while((g=fread(c,1,sizeof c,f))>0) b.insert(b.end(),c,c+g); fclose(f); return b; }
int main(int argc, char** argv){
  std::mt19937 rng(12345);
  long ok=0, fail=0;
  const int iters = argc > 1 ? atoi(argv[1]) : 100;
  for (int a=2;a<argc;++a){
    std::vector<uint8_t> base = readf(argv[a]);
    if (base.empty()) continue;
    for (int it=0; it<iters; ++it){
      std::vector<uint8_t> b = base;
      int kind = rng()%4;
      if (kind==0){ size_t n = rng()% (b.size()+1); b.resize(n); }
      else if (kind==1){ int flips = 1 + rng()%8; for(int k=0;k<flips;++k){ b[rng()%b.size()] ^= (uint8_t)(1u << (rng()%8)); } }
      else if (kind==2){ int m = 1 + rng()%16; for(int k=0;k<m;++k){ b[rng()%b.size()] = (uint8_t)rng(); } }
      else { size_t p = rng()%b.size(); size_t n = rng()%32; for(size_t k=0;k<n && p+k<b.size();++k) b[p+k]=0xff; }
      for (int req=0; req<=4; req+= (it%2?1:4)) {
        Image img; std::string err;
        bool r = decode_image(b.data(), b.size(), req, &img, &err);
        if (r) { ok++; if (img.px.size() != (size_t)img.w*img.h*img.comp) { printf("SIZE MISMATCH %s\n", argv[a]); return 2; } }
        else fail++;
      }
    }
  }
  printf("decoded ok %ld, rejected %ld\n", ok, fail);
  // encoder on random sizes
  for (int it=0; it<200; ++it){ int w=1+rng()%40,h=1+rng()%40,c=1+rng()%4; std::vector<uint8_t> d((size_t)w*h*c); for(auto&v:d) v=(uint8_t)(rng()% (it%3?256:4)); std::vector<uint8_t> png; if(!encode_png(w,h,c,d.data(),0,&png)) return 3; Image img; std::string err; if(!decode_image(png.data(),png.size(),0,&img,&err) || img.px!=d) { printf("ROUNDTRIP FAIL\n"); return 4; } }
  printf("png roundtrips ok\n");
  return 0;
}
