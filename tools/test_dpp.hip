#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include <vector>
#include <cstdlib>
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}

// Cross-lane moves inside a row of 16 lanes (DPP, no LDS round trip):
// 0xB1 = quad_perm[1,0,3,2], 0x4E = quad_perm[2,3,0,1], 0x141 = row_half_mirror, 0x140 = row_mirror.
// Applying them in this order leaves every lane of a row with the row's reduction.
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, kCtrl, 0xf, 0xf, false);
}
template <int kCtrl>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v)
{
    return ((uint64_t)dpp_u32<kCtrl>((uint32_t)(v >> 32)) << 32) | dpp_u32<kCtrl>((uint32_t)v);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l)
{
    // the builtin returns int: cast before widening, or the low half sign-extends
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((uint32_t)v, l);
    return ((uint64_t)hi << 32) | lo;
}

// Wave reductions; the result is wave-uniform (combined from the four rows' lane 0/16/32/48).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, dpp_u32<0xB1>(v));
    v = max(v, dpp_u32<0x4E>(v));
    v = max(v, dpp_u32<0x141>(v));
    v = max(v, dpp_u32<0x140>(v));
    const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += dpp_u32<0xB1>(v);
    v += dpp_u32<0x4E>(v);
    v += dpp_u32<0x141>(v);
    v += dpp_u32<0x140>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v)
{
    v |= dpp_u64<0xB1>(v);
    v |= dpp_u64<0x4E>(v);
    v |= dpp_u64<0x141>(v);
    v |= dpp_u64<0x140>(v);
    return readlane_u64(v, 0) | readlane_u64(v, 16) | readlane_u64(v, 32) | readlane_u64(v, 48);
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    auto mv = [](double x, auto tag) {
        return __longlong_as_double((long long)dpp_u64<decltype(tag)::value>((uint64_t)__double_as_longlong(x)));
    };
    v += mv(v, std::integral_constant<int, 0xB1>{});
    v += mv(v, std::integral_constant<int, 0x4E>{});
    v += mv(v, std::integral_constant<int, 0x141>{});
    v += mv(v, std::integral_constant<int, 0x140>{});
    auto rl = [&](int l) { return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(v), l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}

__global__ void k(const uint32_t* a, const double* d, const uint64_t* o, uint32_t* rmax, uint32_t* rsum, double* rd, uint64_t* ror) {
  int l = threadIdx.x;
  uint32_t m = wave_max_u32(a[l]); uint32_t s = wave_sum_u32(a[l] & 0xffff); double ds = wave_sum_f64(d[l]); uint64_t r = wave_or_u64(o[l]);
  rmax[l]=m; rsum[l]=s; rd[l]=ds; ror[l]=r;
}
int main(){
  int bad=0;
  for(int t=0;t<200;t++){
    std::vector<uint32_t> a(64); std::vector<double> d(64); std::vector<uint64_t> o(64);
    uint32_t em=0, es=0; double ed=0; uint64_t eo=0;
    for(int i=0;i<64;i++){ a[i]=(t%3==0 && i>14)?0:(uint32_t)rand()*2654435761u; d[i]=(double)rand()/RAND_MAX; o[i]=(rand()%8==0)?(1ull<<(rand()%64)):0; em=a[i]>em?a[i]:em; es+=a[i]&0xffff; ed+=d[i]; eo|=o[i]; }
    uint32_t *da,*rm,*rs; double *dd,*rd; uint64_t *dor,*ror;
    hipMalloc(&da,256); hipMalloc(&rm,256); hipMalloc(&rs,256); hipMalloc(&dd,512); hipMalloc(&rd,512); hipMalloc(&dor,512); hipMalloc(&ror,512);
    hipMemcpy(da,a.data(),256,hipMemcpyHostToDevice); hipMemcpy(dd,d.data(),512,hipMemcpyHostToDevice); hipMemcpy(dor,o.data(),512,hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,da,dd,dor,rm,rs,rd,ror);
    std::vector<uint32_t> hm(64),hs(64); std::vector<double> hd(64); std::vector<uint64_t> ho(64);
    hipMemcpy(hm.data(),rm,256,hipMemcpyDeviceToHost); hipMemcpy(hs.data(),rs,256,hipMemcpyDeviceToHost); hipMemcpy(hd.data(),rd,512,hipMemcpyDeviceToHost); hipMemcpy(ho.data(),ror,512,hipMemcpyDeviceToHost);
    for(int i=0;i<64;i++){ if(hm[i]!=em||hs[i]!=es||ho[i]!=eo||fabs(hd[i]-ed)>1e-9){ if(bad<5) printf("t=%d lane %d max %u/%u sum %u/%u or %llx/%llx d %f/%f\n",t,i,hm[i],em,hs[i],es,(unsigned long long)ho[i],(unsigned long long)eo,hd[i],ed); bad++; } }
    hipFree(da);hipFree(rm);hipFree(rs);hipFree(dd);hipFree(rd);hipFree(dor);hipFree(ror);
  }
  printf("bad=%d\n",bad); return bad!=0;
}
