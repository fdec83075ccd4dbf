// pcd_fuzz.cpp -- mutation fuzzing of gpx_pcd_read (untrusted input files) under AddressSanitizer / UBSan on the CPU:
// byte flips, truncation, corrupted header numbers, absurd POINTS counts, garbage bodies.  Built and run by
// tests/test_host.py.   usage: pcd_fuzz <iterations per file> <file.pcd>...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include "gpx.h"
static std::vector<unsigned char> slurp(const char *p){FILE*f=fopen(p,"rb");std::vector<unsigned char>b;if(!f)return b;fseek(f,0,SEEK_END);long n=ftell(f);fseek(f,0,SEEK_SET);b.resize(n);if(fread(b.data(),1,n,f)!=(size_t)n)b.clear();fclose(f);return b;}
int main(int argc,char**argv){
  std::mt19937 rng(12345);
  const char *tmpname = getenv("PCD_FUZZ_TMP") ? getenv("PCD_FUZZ_TMP") : "/tmp/gpx_fuzz.pcd";
  int total=0, ok=0;
  const int iters=argc>1?atoi(argv[1]):100;
  for(int a=2;a<argc;++a){
    std::vector<unsigned char> orig=slurp(argv[a]);
    if(orig.empty())continue;
    for(int it=0;it<iters;++it){
      std::vector<unsigned char> b=orig;
      int mode=rng()%5;
      if(mode==0){ // flip random bytes
        int k=1+rng()%8; for(int i=0;i<k;++i) b[rng()%b.size()]=(unsigned char)rng();
      } else if(mode==1){ // truncate
        b.resize(rng()%b.size());
      } else if(mode==2){ // corrupt header numbers
        size_t lim=std::min<size_t>(b.size(),300); for(int i=0;i<6;++i){size_t p=rng()%lim; if(b[p]>='0'&&b[p]<='9') b[p]='0'+rng()%10;}
      } else if(mode==3){ // huge POINTS / WIDTH
        std::string s((char*)b.data(), std::min<size_t>(b.size(),400)); size_t p=s.find("POINTS "); if(p!=std::string::npos){ const char*big="POINTS 4000000000"; for(size_t i=0;i<strlen(big)&&p+i<b.size();++i) b[p+i]=big[i]; }
      } else { // random garbage in the body
        for(size_t i=b.size()/2;i<b.size();i+=1+rng()%7) b[i]=(unsigned char)rng();
      }
      FILE*f=fopen(tmpname,"wb"); fwrite(b.data(),1,b.size(),f); fclose(f);
      long n=gpx_pcd_read(tmpname,nullptr,0);
      ++total;
      if(n>0 && n<50000000){ std::vector<float> xyz(3*(size_t)n); long n2=gpx_pcd_read(tmpname,xyz.data(),(size_t)n); if(n2==n)++ok; }
    }
  }
  printf("fuzzed %d inputs, %d still decodable, no crash\n",total,ok);
  return 0;
}
