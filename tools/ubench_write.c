#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/mman.h>
#include <time.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
typedef struct { int fd; unsigned char *dst; const unsigned char *src; size_t n; off_t off; int mode; } job;
static void *run(void *a){ job *j=a; if(j->mode==0){ size_t d=0; while(d<j->n){ ssize_t w=pwrite(j->fd,j->src+d,j->n-d,j->off+d); if(w<=0) break; d+=w;} } else memcpy(j->dst+j->off,j->src,j->n); return NULL; }
int main(int argc,char**argv){
  const char *path=argv[1]; int mode=atoi(argv[2]); int T=atoi(argv[3]); size_t total=(size_t)atol(argv[4])<<20, chunk=64u<<20;
  unsigned char *buf=malloc(chunk); memset(buf,7,chunk);
  int fd=open(path,O_RDWR|O_CREAT|O_TRUNC,0644); unsigned char *map=NULL;
  double t0=now();
  if(mode==1){ if(ftruncate(fd,total)) return 1; map=mmap(NULL,total,PROT_READ|PROT_WRITE,MAP_SHARED,fd,0); if(map==MAP_FAILED){perror("mmap");return 1;} }
  for(size_t o=0;o<total;o+=chunk){ pthread_t th[64]; job jb[64];
    for(int i=0;i<T;i++){ size_t lo=chunk*i/T, hi=chunk*(i+1)/T; jb[i]=(job){fd,map,buf+lo,hi-lo,(off_t)(o+lo),mode}; if(i+1<T) pthread_create(&th[i],NULL,run,&jb[i]); }
    run(&jb[T-1]); for(int i=0;i+1<T;i++) pthread_join(th[i],NULL); }
  if(map) munmap(map,total);
  close(fd);
  double dt=now()-t0; printf("mode %d T %d: %.0f ms  %.2f GB/s\n",mode,T,dt*1e3,total/dt/1e9); unlink(path); return 0; }
