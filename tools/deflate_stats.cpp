// DEFLATE symbol statistics of a BGZF file (literals / matches, match lengths and distances): g++ -O2 -std=c++17 -o deflate_stats tools/deflate_stats.cpp; ./deflate_stats x.bam
// Test-side tool: it runs the decoder core of secphase_amd/csrc/spx_inflate.h on the host with a counting environment.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "../secphase_amd/csrc/spx_inflate.h"
using namespace spxz;
struct CEnv : HostEnvT<9,8> {
    uint64_t nlit=0,nmatch=0,mlen=0,hist_len[300]={0},near256=0,near512=0,near1k=0,near2k=0,overlap=0, bytes_far=0, bytes_match=0;
    void lit_push(uint8_t c){ nlit++; HostEnvT<9,8>::lit_push(c);}    
    bool put_literal(uint8_t c){ nlit++; return HostEnvT<9,8>::put_literal(c);}    
    void copy_match(int len,int dist){ nmatch++; mlen+=len; hist_len[len]++; if(dist<=256)near256++; if(dist<=512)near512++; if(dist<=1024)near1k++; if(dist<=2048)near2k++; if(dist<len)overlap++; if(dist>512)bytes_far+=len; bytes_match+=len; HostEnvT<9,8>::copy_match(len,dist);}    
};
int main(int argc,char**argv){
    FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); std::vector<uint8_t> b(n); fread(b.data(),1,n,f);
    std::vector<uint8_t> out(70000);
    CEnv tot; uint64_t NL=0,NM=0,ML=0,n256=0,n512=0,n1k=0,n2k=0,ov=0,bf=0,bm=0,outb=0; uint64_t hl[300]={0}; long nb=0;
    long at=0; while(at<n){ int bs=(b[at+16]|(b[at+17]<<8))+1; int xlen=b[at+10]|(b[at+11]<<8); const uint8_t*d=&b[at+12+xlen]; int clen=bs-xlen-12-8; uint32_t isz; memcpy(&isz,&b[at+bs-4],4);
        CEnv e; e.in=d; e.in_len=clen; e.out=out.data(); e.cap=65536; int rc=inflate_stream(e,(int64_t)clen*8,isz); if(rc){printf("rc %d\n",rc);return 1;}
        NL+=e.nlit;NM+=e.nmatch;ML+=e.mlen;n256+=e.near256;n512+=e.near512;n1k+=e.near1k;n2k+=e.near2k;ov+=e.overlap;bf+=e.bytes_far;bm+=e.bytes_match;outb+=isz; for(int k=0;k<300;k++)hl[k]+=e.hist_len[k]; nb++; at+=bs; }
    printf("blocks %ld out %.1f MB; literals %lu matches %lu (%.1f%% of symbols), avg len %.1f, bytes/symbol %.2f\n",nb,outb/1e6,NL,NM,100.0*NM/(NL+NM),(double)ML/NM,(double)outb/(NL+NM));
    printf("dist<=256 %.1f%% <=512 %.1f%% <=1k %.1f%% <=2k %.1f%% overlap %.1f%%; bytes from matches %.1f%%, from far(>512) matches %.1f%%\n",100.0*n256/NM,100.0*n512/NM,100.0*n1k/NM,100.0*n2k/NM,100.0*ov/NM,100.0*bm/outb,100.0*bf/outb);
    uint64_t c=0; for(int k=3;k<=258;k++){c+=hl[k]; if(k==4||k==8||k==16||k==32||k==64||k==128||k==257||k==258)printf("len<=%d %.1f%%  ",k,100.0*c/NM);} printf("\n");
}
