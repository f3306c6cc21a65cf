// Mutation fuzzer for the host-side parsers (DSF / DFF headers, ID3v2 tags): they read untrusted files.
// Build with -fsanitize=address,undefined and run:  fuzz_host_parsers <seed file .dsf|.dff> <seed> <iterations> <tmp dir>
// (tests/test_host_parsers_fuzz.py does that with synthetic seed files)
#include "../dsd2dxd_amd/csrc/host/dsd_reader.h"
#include "../dsd2dxd_amd/csrc/host/id3_tag.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
using namespace d2dhost;
static std::vector<uint8_t> slurp(const char* p){ FILE*f=fopen(p,"rb"); std::vector<uint8_t> b; if(!f) return b; fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); b.resize(n); fread(b.data(),1,n,f); fclose(f); return b; }
int main(int argc,char**argv){
  unsigned seed=argc>2?atoi(argv[2]):1; int iters=argc>3?atoi(argv[3]):2000;
  std::vector<uint8_t> base=slurp(argv[1]);
  const char* ext = strrchr(argv[1],'.');
  std::string tmp=std::string(std::string(argc > 4 ? argv[4] : "/tmp") + "/d2d_fuzz_m")+ext;
  unsigned x=seed; auto rnd=[&]{ x=x*1664525u+1013904223u; return x>>8; };
  {  // the seed file as it is (regression files are fed this way, with 0 iterations)
    DsdInfo info; std::string e=probe(argv[1],info);
    if(e.empty()){ std::vector<uint8_t> tag; std::string w; read_source_tag(argv[1],info,tag,w);
      DsdSource src; if(src.open(argv[1],info).empty()){ std::vector<uint8_t> buf(8192*std::max(1u,info.channels>64?64u:info.channels)); for(int q=0;q<4;++q){ long n=src.read(buf.data(),8192); if(n<=0) break; } } }
  }
  for(int it=0;it<iters;++it){
    std::vector<uint8_t> b=base;
    int nm=1+rnd()%8;
    for(int k=0;k<nm;++k){ size_t pos; unsigned r=rnd()%10; if(r<6) pos=rnd()%std::min<size_t>(b.size(),160); else if(r<9) pos=b.size()-1-rnd()%std::min<size_t>(b.size(),1300); else pos=rnd()%b.size(); b[pos]=(uint8_t)rnd(); }
    if(rnd()%5==0) b.resize(rnd()%b.size());
    FILE*f=fopen(tmp.c_str(),"wb"); fwrite(b.data(),1,b.size(),f); fclose(f);
    DsdInfo info; std::string e=probe(tmp,info);
    if(e.empty()){
      std::vector<uint8_t> tag; std::string w; read_source_tag(tmp,info,tag,w);
      if(!tag.empty()){ append_to_album(tag, " [88.2K]"); std::vector<std::pair<std::string,std::string>> f2; std::vector<TagPicture> p2; tag_to_vorbis(tag,f2,p2); }
      DsdSource src; if(src.open(tmp,info).empty()){ std::vector<uint8_t> buf(8192*std::max(1u,info.channels>64?64u:info.channels)); for(int q=0;q<4;++q){ long n=src.read(buf.data(),8192); if(n<=0) break; } }
    }
  }
  // direct tag fuzz
  for(int it=0;it<iters*5;++it){
    std::vector<uint8_t> t={'I','D','3',(uint8_t)(2+rnd()%3),0,(uint8_t)(rnd()%4?0:rnd()),0,0,(uint8_t)(rnd()%3),(uint8_t)(rnd()&127)};
    size_t n=rnd()%300; for(size_t i=0;i<n;++i){ unsigned r=rnd()%8; t.push_back(r==0?'T':r==1?'A':r==2?0:(uint8_t)rnd()); }
    if(rnd()%2){ const char* ids[]={"TALB","TIT2","APIC","COMM","TRCK","TAL","PIC"}; const char* id=ids[rnd()%7]; size_t at=10; if(t.size()>at+12){ memcpy(&t[at],id,strlen(id)); t[at+4]=0;t[at+5]=0;t[at+6]=0;t[at+7]=(uint8_t)(rnd()%64); } }
    append_to_album(t," [96K]"); std::vector<std::pair<std::string,std::string>> f2; std::vector<TagPicture> p2; tag_to_vorbis(t,f2,p2);
  }
  puts("fuzz ok"); return 0; }
