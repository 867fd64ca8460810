// Robustness harness (CPU, ASAN/UBSAN): corrupted containers must come back as errors or decode to
// something -- never crash or read out of bounds.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <random>
#include <vector>
#include "vorbispizza_front.h"
static std::vector<uint8_t> slurp(const char *p){ FILE*f=fopen(p,"rb"); std::vector<uint8_t> d; if(!f) return d; fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); d.resize(n); if(fread(d.data(),1,n,f)!=(size_t)n) d.clear(); fclose(f); return d; }
static uint32_t crc_tab[256];
static void crc_init(){ for(uint32_t i=0;i<256;++i){ uint32_t r=i<<24; for(int k=0;k<8;++k) r=(r&0x80000000u)?((r<<1)^0x04C11DB7u):(r<<1); crc_tab[i]=r; } }
// walk the pages and give each one a valid checksum again, so that mutations reach the bit-level decoder
static void fix_crcs(std::vector<uint8_t>&d){
  size_t pos=0;
  while(pos+27<=d.size()){
    if(!(d[pos]=='O'&&d[pos+1]=='g'&&d[pos+2]=='g'&&d[pos+3]=='S')){ ++pos; continue; }
    int nseg=d[pos+26]; if(pos+27+nseg>d.size()) break;
    size_t body=0; for(int i=0;i<nseg;++i) body+=d[pos+27+i];
    size_t len=27+nseg+body; if(pos+len>d.size()) break;
    d[pos+22]=d[pos+23]=d[pos+24]=d[pos+25]=0;
    uint32_t c=0; for(size_t i=0;i<len;++i) c=(c<<8)^crc_tab[((c>>24)&0xFF)^d[pos+i]];
    d[pos+22]=c&0xFF; d[pos+23]=(c>>8)&0xFF; d[pos+24]=(c>>16)&0xFF; d[pos+25]=(c>>24)&0xFF;
    pos+=len;
  }
}
static int run(const std::vector<uint8_t>&d){
  vpzh_stream*s=nullptr; int rc=vpzh_open_memory(d.data(),d.size(),&s);
  if(rc!=0){ if(s) vpzh_close(s); return 1; }
  vpzh_info info; vpzh_get_info(s,&info);
  if(info.audio_packets>0 && info.channels>0){
    std::vector<vpz_packet> pk(info.audio_packets); std::vector<float> res(info.residue_floats+1); std::vector<int16_t> posts((size_t)info.audio_packets*info.channels*64); std::vector<uint8_t> cnt((size_t)info.audio_packets*info.channels);
    int st=vpzh_max_floor0_order(s); std::vector<float> amp((size_t)info.audio_packets*info.channels+1), co((size_t)info.audio_packets*info.channels*(st>0?st:1)+1);
    int64_t used=0;
    rc=vpzh_decode_range_ex(s,0,info.audio_packets,0,0,pk.data(),res.data(),posts.data(),cnt.data(),&used,st?amp.data():nullptr,st?co.data():nullptr,st);
    // (ABI v5) the int16 form of the same decode, where the -- possibly mutated -- setup header still promises integers: the float
    // values must be those integers, packet for packet
    if(rc==0 && vpzh_residue_is_integral(s)){
      std::vector<vpz_packet> pk2(info.audio_packets); std::vector<int16_t> res16(info.residue_floats+1); int64_t used2=0;
      int rc2=vpzh_decode_range_i16(s,0,info.audio_packets,0,0,pk2.data(),res16.data(),posts.data(),cnt.data(),&used2,st?amp.data():nullptr,st?co.data():nullptr,st);
      if(rc2!=0 || used2!=used){ fprintf(stderr,"int16 decode differs in extent\n"); abort(); }
      for(int64_t k=0;k<info.audio_packets;++k){
        if((pk[k].flags&VPZ_PKT_NOT_DECODED)!=(pk2[k].flags&VPZ_PKT_NOT_DECODED)){ fprintf(stderr,"int16 decode differs in a packet's fate\n"); abort(); }
      }
      for(int64_t i=0;i<used;++i) if((float)res16[i]!=res[i]){ fprintf(stderr,"int16 residue differs at %lld: %d vs %g\n",(long long)i,(int)res16[i],(double)res[i]); abort(); }
    }
    int64_t a,b; vpzh_seek(s, vpzh_total_samples(s)/2, &a,&b);
  }
  vpzh_close(s); return rc?2:0;
}
int main(int argc,char**argv){
  crc_init(); int iters=atoi(argv[1]); std::mt19937 rng(getenv("FUZZ_SEED") ? (unsigned)atoi(getenv("FUZZ_SEED")) : 1234u); int ok=0,err_open=0,err_dec=0;
  for(int f=2; f<argc; ++f){
    std::vector<uint8_t> orig=slurp(argv[f]); if(orig.empty()) continue;
    run(orig);
    for(int it=0; it<iters; ++it){
      std::vector<uint8_t> d=orig;
      int kind=rng()%4;
      int nmut=1+rng()%8;
      for(int m=0;m<nmut;++m){
        size_t pos = (kind==0)? rng()%std::min<size_t>(d.size(),6000) : rng()%d.size();   // kind 0: hit the headers
        if(kind==3){ d.resize(pos+1); break; }       // truncation
        d[pos] = (kind==2)? (uint8_t)rng() : (uint8_t)(d[pos]^(1u<<(rng()%8)));
      }
      if(rng()%4) fix_crcs(d);
      if(getenv("FUZZ_VERBOSE")){ fprintf(stderr,"%s #%d\n",argv[f],it); fflush(stderr); }  // (which case a run hangs in)
      if(const char*dump=getenv("FUZZ_DUMP_CASE")){ int wf=0,wi=0; if(sscanf(dump,"%d:%d",&wf,&wi)==2 && wf==f && wi==it){ FILE*o=fopen("fuzz_case.ogg","wb"); if(o){ fwrite(d.data(),1,d.size(),o); fclose(o);} } }
      int r=run(d); if(r==0)++ok; else if(r==1)++err_open; else ++err_dec;
    }
  }
  printf("fuzz: decoded %d, rejected at open %d, rejected at decode %d\n",ok,err_open,err_dec);
}
