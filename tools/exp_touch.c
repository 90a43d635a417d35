#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>
static const unsigned char *base; static size_t size; static int nt; static int mode;
static void *w(void *a){ long t=(long)a; size_t lo=size/nt*t, hi=t+1==nt?size:size/nt*(t+1); unsigned long s=0;
 if(mode==0){ for(size_t i=lo;i<hi;i+=4096) s+=base[i]; }
 else if(mode==1){ const unsigned char*p=base+lo,*e=base+hi; while(p<e){ const unsigned char*n=memchr(p,'\n',e-p); if(!n)break; s++; p=n+1;} }
 else { for(size_t i=lo;i<hi;i+=64) s+=base[i]; }
 return (void*)s; }
int main(int c,char**v){ int fd=open(v[1],O_RDONLY); struct stat st; fstat(fd,&st); size=st.st_size; nt=atoi(v[2]); mode=atoi(v[3]);
 struct timespec a,b; clock_gettime(CLOCK_MONOTONIC,&a);
 base=mmap(NULL,size,PROT_READ,MAP_PRIVATE,fd,0); madvise((void*)base,size,MADV_SEQUENTIAL|MADV_WILLNEED);
 pthread_t th[64]; for(long t=0;t<nt;t++)pthread_create(&th[t],0,w,(void*)t); unsigned long s=0; for(int t=0;t<nt;t++){void*r;pthread_join(th[t],&r);s+=(unsigned long)r;}
 clock_gettime(CLOCK_MONOTONIC,&b); printf("mode %d threads %d: %.3f s (sum %lu)\n",mode,nt,(b.tv_sec-a.tv_sec)+(b.tv_nsec-a.tv_nsec)*1e-9,s); }
