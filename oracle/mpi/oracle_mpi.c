/*
 * oracle_mpi.c -- TEST / BASELINE INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The halo exchange of the CPU baseline: mp_exchange2d/3d/4d of the reference
 * (ROMS/Utility/mp_exchange.F:290/1413/2753) with its neighbour table
 * (tile_neighbors, :73-286) restated in C on MPI -- two dependent phases, W/E
 * first, then S/N over the full i-range so that the corners ride along,
 * MPI_Irecv / MPI_Isend / MPI_Waitall on packed buffers, exactly the call
 * pattern of the reference.  bench.py's `cpu_baseline` leg runs one oracle
 * process per tile under mpiexec on all host cores of the GPU box and installs
 * oracle_mpi_exchange as the oracle's exchange hook (oracle.h:
 * oracle_set_exchange_hook), so the baseline is "the port, tiled and exchanged
 * the way the reference's MPI build is".
 */
#include <mpi.h>
#include <stdlib.h>
#include <string.h>
#include "roms_hip.h"

static roms_bounds_t B;
static int g_rank = 0, g_size = 1, g_we_inited = 0;
static int Wtile, Etile, Stile, Ntile;
static int GsendW, GsendE, GrecvW, GrecvE, GsendS, GsendN, GrecvS, GrecvN;
static double *g_buf[4] = {0, 0, 0, 0};
static size_t g_cap = 0;

int oracle_mpi_init(void)
{
  int flag = 0;
  MPI_Initialized(&flag);
  if (!flag) { MPI_Init(0, 0); g_we_inited = 1; }
  MPI_Comm_rank(MPI_COMM_WORLD, &g_rank);
  MPI_Comm_size(MPI_COMM_WORLD, &g_size);
  return g_rank;
}
int oracle_mpi_size(void) { return g_size; }
void oracle_mpi_barrier(void) { MPI_Barrier(MPI_COMM_WORLD); }
double oracle_mpi_wtime(void) { return MPI_Wtime(); }
double oracle_mpi_max(double x)
{
  double y = x;
  MPI_Allreduce(&x, &y, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
  return y;
}
/* diag.F:398-420: the tile-local results meet on every rank (mp_reduce / mp_reduce2) */
void oracle_mpi_allgather12(const double *in12, double *out)
{
  MPI_Allgather((void *)in12, 12, MPI_DOUBLE, out, 12, MPI_DOUBLE, MPI_COMM_WORLD);
}
void oracle_mpi_finalize(void)
{
  for (int q = 0; q < 4; q++) { free(g_buf[q]); g_buf[q] = 0; }
  g_cap = 0;
  if (g_we_inited) MPI_Finalize();
  g_we_inited = 0;
}

static int table(int i, int j)
{
  return (i < 0 || i >= B.ntileI || j < 0 || j >= B.ntileJ) ? -1 : j * B.ntileI + i;
}

/* tile_neighbors, mp_exchange.F:73-286 */
void oracle_mpi_setup(const roms_bounds_t *b)
{
  B = *b;
  const int Ng = B.NghostPoints;
  const int I = g_rank % B.ntileI, J = g_rank / B.ntileI;
  GsendW = GsendE = GrecvW = GrecvE = GsendS = GsendN = GrecvS = GrecvN = Ng;
  Wtile = table(I - 1, J);
  Etile = table(I + 1, J);
  if (B.EWperiodic && B.ntileI > 1) {
    if (table(I - 1, J) < 0) { Wtile = table(B.ntileI - 1, J); if (Ng != 3) GrecvW = Ng + 1; }
    else if (table(I + 1, J) < 0) { Etile = table(0, J); if (Ng != 3) GsendE = Ng + 1; }
  }
  Stile = table(I, J - 1);
  Ntile = table(I, J + 1);
  if (B.NSperiodic && B.ntileJ > 1) {
    if (table(I, J - 1) < 0) { Stile = table(I, B.ntileJ - 1); if (Ng != 3) GrecvS = Ng + 1; }
    else if (table(I, J + 1) < 0) { Ntile = table(I, 0); if (Ng != 3) GsendN = Ng + 1; }
  }
}

static void ensure(size_t n)
{
  if (n <= g_cap) return;
  for (int q = 0; q < 4; q++) { free(g_buf[q]); g_buf[q] = (double *)malloc(n * sizeof(double)); }
  g_cap = n;
}

/* copy the block [i0,i0+wi) x [j0,j0+wj) of all nk planes to / from a packed buffer */
static void pack(double *A, int nk, int i0, int wi, int j0, int wj, double *buf, int unpack)
{
  const long ni = B.UBi - B.LBi + 1, nij = ni * (long)(B.UBj - B.LBj + 1);
  long q = 0;
  for (int k = 0; k < nk; k++)
    for (int j = j0; j < j0 + wj; j++) {
      double *row = A + (long)(i0 - B.LBi) + (long)(j - B.LBj) * ni + (long)k * nij;
      if (unpack) memcpy(row, buf + q, sizeof(double) * wi);
      else memcpy(buf + q, row, sizeof(double) * wi);
      q += wi;
    }
}

/* the exchange hook: mp_exchange2d (nk = 1) / mp_exchange3d / mp_exchange4d (nk = N * NT) */
void oracle_mpi_exchange(double *A, int nk, int gtype)
{
  (void)gtype;
  const int nj = B.UBj - B.LBj + 1, ni = B.UBi - B.LBi + 1;
  MPI_Request req[4];
  MPI_Status stat[4];
  int nreq;
  /* ---- phase 1: western and eastern edges, every row of the array (mp_exchange.F:395-560) ---- */
  if (Wtile >= 0 || Etile >= 0) {
    ensure((size_t)nk * nj * (B.NghostPoints + 1));
    nreq = 0;
    if (Wtile >= 0) MPI_Irecv(g_buf[0], nk * nj * GrecvW, MPI_DOUBLE, Wtile, 2, MPI_COMM_WORLD, &req[nreq++]);
    if (Etile >= 0) MPI_Irecv(g_buf[1], nk * nj * GrecvE, MPI_DOUBLE, Etile, 1, MPI_COMM_WORLD, &req[nreq++]);
    if (Wtile >= 0) {
      pack(A, nk, B.Istr, GsendW, B.LBj, nj, g_buf[2], 0);
      MPI_Isend(g_buf[2], nk * nj * GsendW, MPI_DOUBLE, Wtile, 1, MPI_COMM_WORLD, &req[nreq++]);
    }
    if (Etile >= 0) {
      pack(A, nk, B.Iend - GsendE + 1, GsendE, B.LBj, nj, g_buf[3], 0);
      MPI_Isend(g_buf[3], nk * nj * GsendE, MPI_DOUBLE, Etile, 2, MPI_COMM_WORLD, &req[nreq++]);
    }
    MPI_Waitall(nreq, req, stat);        /* (MPI_STATUSES_IGNORE trips gcc's -Wstringop-overflow with this mpi.h) */
    if (Wtile >= 0) pack(A, nk, B.Istr - GrecvW, GrecvW, B.LBj, nj, g_buf[0], 1);
    if (Etile >= 0) pack(A, nk, B.Iend + 1, GrecvE, B.LBj, nj, g_buf[1], 1);
  }
  /* ---- phase 2: southern and northern edges over the full i-range (:562-730) ---- */
  if (Stile >= 0 || Ntile >= 0) {
    ensure((size_t)nk * ni * (B.NghostPoints + 1));
    nreq = 0;
    if (Stile >= 0) MPI_Irecv(g_buf[0], nk * ni * GrecvS, MPI_DOUBLE, Stile, 4, MPI_COMM_WORLD, &req[nreq++]);
    if (Ntile >= 0) MPI_Irecv(g_buf[1], nk * ni * GrecvN, MPI_DOUBLE, Ntile, 3, MPI_COMM_WORLD, &req[nreq++]);
    if (Stile >= 0) {
      pack(A, nk, B.LBi, ni, B.Jstr, GsendS, g_buf[2], 0);
      MPI_Isend(g_buf[2], nk * ni * GsendS, MPI_DOUBLE, Stile, 3, MPI_COMM_WORLD, &req[nreq++]);
    }
    if (Ntile >= 0) {
      pack(A, nk, B.LBi, ni, B.Jend - GsendN + 1, GsendN, g_buf[3], 0);
      MPI_Isend(g_buf[3], nk * ni * GsendN, MPI_DOUBLE, Ntile, 4, MPI_COMM_WORLD, &req[nreq++]);
    }
    MPI_Waitall(nreq, req, stat);        /* (MPI_STATUSES_IGNORE trips gcc's -Wstringop-overflow with this mpi.h) */
    if (Stile >= 0) pack(A, nk, B.LBi, ni, B.Jstr - GrecvS, GrecvS, g_buf[0], 1);
    if (Ntile >= 0) pack(A, nk, B.LBi, ni, B.Jend + 1, GrecvN, g_buf[1], 1);
  }
}
