// kernels_upload.h -- Device kernels, part 1a: upload (column means, pack into the resident layout) and the incomplete-row side tables.
// Device code shared by the translation units of libplspm_hip.so (host_internal.h lists them); not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------------ upload kernels
// Column sums, row-major source: block = 64 columns x 4 row lanes; partial[blockIdx.x][p].
__global__ void __launch_bounds__(256) colsum_rowmajor_kernel(const double* __restrict__ X, long N, int src_cols, const int* __restrict__ colidx,
                                                               int P, double* __restrict__ partial) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = lmin(N, r0 + rows_per_block);
    for (int pbase = 0; pbase < P; pbase += 64) {
        const int p = pbase + tx;
        double s = 0.0;
        if (p < P) {
            const int c = colidx[p];
            for (long i = r0 + ty; i < r1; i += 4) s += X[i * src_cols + c];
        }
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && p < P) partial[(long)blockIdx.x * P + p] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        __syncthreads();
    }
}
// Column sums, column-major source: grid (chunks, P); threads run along the rows.
__global__ void __launch_bounds__(256) colsum_colmajor_kernel(const double* __restrict__ X, long N, const int* __restrict__ colidx, int P,
                                                               double* __restrict__ partial) {
    __shared__ double red[256];
    const int p = blockIdx.y;
    const double* col = X + (long)colidx[p] * N;
    const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = lmin(N, r0 + rows_per_block);
    double s = 0.0;
    for (long i = r0 + threadIdx.x; i < r1; i += 256) s += col[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h]; __syncthreads(); }
    if (threadIdx.x == 0) partial[(long)blockIdx.x * P + p] = red[0];
}
__global__ void colmean_kernel(const double* __restrict__ partial, int nblk, int P, long N, double* __restrict__ shift) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(long)b * P + p];
    shift[p] = s / (double)N;
}
// Xa[i][p] = X[i][colidx[p]] - shift[p] (p < P), 1 (p == P), 0 (p > P).  Row-major source: one thread per output element.
__global__ void __launch_bounds__(256) pack_rowmajor_kernel(const double* __restrict__ X, long N, int src_cols, const int* __restrict__ colidx,
                                                             int P, int PA, const double* __restrict__ shift, double* __restrict__ Xa) {
    const long total = N * PA;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long i = e / PA;
        const int p = (int)(e - i * PA);
        double v = 0.0;
        if (p < P) v = X[i * src_cols + colidx[p]] - shift[p];
        else if (p == P) v = 1.0;
        Xa[e] = v;
    }
}
// Column-major source: 64-row x 32-column LDS transpose tile (reads run along rows, writes along columns).
__global__ void __launch_bounds__(256) pack_colmajor_kernel(const double* __restrict__ X, long N, const int* __restrict__ colidx, int P, int PA,
                                                             const double* __restrict__ shift, double* __restrict__ Xa) {
    __shared__ double tile[32][65];
    const long i0 = (long)blockIdx.x * 64;
    const int p0 = blockIdx.y * 32;
    {
        const int r = threadIdx.x & 63;
        for (int c = threadIdx.x >> 6; c < 32; c += 4) {
            const int p = p0 + c;
            const long i = i0 + r;
            double v = 0.0;
            if (i < N) {
                if (p < P) v = X[(long)colidx[p] * N + i] - shift[p];
                else if (p == P) v = 1.0;
            }
            tile[c][r] = v;
        }
    }
    __syncthreads();
    {
        const int c = threadIdx.x & 31;
        for (int r = threadIdx.x >> 5; r < 64; r += 8) {
            const long i = i0 + r;
            if (i < N && p0 + c < PA) Xa[i * PA + p0 + c] = tile[c][r];
        }
    }
}

// plspm_model_set_incomplete_rows: copy the incomplete rows (masked) into the side tables and zero them in Xa (data + ones column)
__global__ void __launch_bounds__(256) extract_rows_kernel(double* __restrict__ Xa, int PA, int P, const int* __restrict__ rowid, const unsigned char* __restrict__ mask,
                                                           double* __restrict__ Xk, double* __restrict__ Mk) {
    const long j = blockIdx.x;
    double* row = Xa + (long)rowid[j] * PA;
    for (int p = threadIdx.x; p < PA; p += blockDim.x) {
        if (p < P) {
            const double present = mask[j * P + p] ? 1.0 : 0.0;
            Mk[j * P + p] = present;
            Xk[j * P + p] = present * row[p];
        }
        row[p] = 0.0;
    }
}

