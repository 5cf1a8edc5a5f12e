/* rsrec.h -- C ABI of librsrec: the MI355X (gfx950) recursion engine for RS-LMTO-ASA.
 *
 * Drop-in boundary for ONE hot path of rslmtoasa/rslmtoasa: the Haydock / block-Lanczos /
 * Chebyshev recursion of source/recursion.f90.  The reference has no FFI today (it is 100 %
 * Fortran); these entry points are what a `bind(C)` interface block in a replacement
 * `recursion_mod` binds (see fortran/rsrec_binding.f90 and INTEGRATION.md).  Each entry point
 * cites the reference procedure it replaces.
 *
 * Conventions (all follow the reference so Fortran arrays are passed as they are):
 *   - every array is caller-allocated HOST memory in Fortran (column-major) order;
 *   - complex(8) data are passed as `const double*` pointing at interleaved (re,im) pairs,
 *     i.e. exactly Fortran `complex(rp)` / C `double _Complex` storage;
 *   - atom numbers are 1-based, 0 = "no neighbour" (lattice.f90:1854, nn(kk, nnmax+1));
 *   - every function returns 0 on success, non-zero on error; the message is read with
 *     rsrec_last_error().  The Fortran shim turns non-zero into g_logger%fatal, which is the
 *     reference's only error behaviour on this path (recursion.f90:1942, :2595).
 *   - one handle per process/GPU; not re-entrant (like the reference's `this%` scratch).
 *   - there is NO CPU fallback: rsrec_create fails if no gfx950 device is usable.
 */
#ifndef RSREC_H
#define RSREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsrec_handle rsrec_t;

#define RSREC_OK 0
#define RSREC_ERR_ARG 1        /* bad argument / call order */
#define RSREC_ERR_DEVICE 2     /* HIP runtime error (no device, out of memory, launch failure) */
#define RSREC_ERR_DIVERGED 3   /* Chebyshev moments blew up: recursion.f90:2594-2596 (fatal in the reference) */
#define RSREC_ERR_EIG 4        /* 18x18 eigen-solve did not converge: recursion.f90:1942 (zheev info /= 0) */

/* Library/ABI version (major*100 + minor). */
int rsrec_version(void);

/* Number of usable HIP devices (0 if none): the Fortran host maps MPI rank -> device with it (mpi.f90 rank). */
int rsrec_device_count(void);

/* Create / destroy the per-GPU context.  `device` = HIP device ordinal (the rank's local GPU).
 * Replaces nothing in the reference (its recursion type owns host arrays only, recursion.f90:41-116,
 * allocated in restore_to_default :3713-3790); the drop-in module creates the context lazily on the
 * first recur* call (finalizer trap: SURVEY.md section 8b). */
int rsrec_create(rsrec_t **out, int device);
int rsrec_destroy(rsrec_t *h);

/* Geometry tables read by the recursion: lattice%kk, %nn, %iz, %nmax, %ntype
 * (lattice.f90:138-239; read at recursion.f90:1577-1580, :1603-1606).
 *   nn  : int32 (kk, nncols) column-major; nn(i,1) = number of slots incl. on-site, nn(i,j>=2) = atom or 0
 *   iz  : int32 (kk) atom type, 1..ntype
 *   nmax: the first nmax atoms carry per-atom blocks `hall` (impurity region), 0 for bulk/surface */
int rsrec_set_lattice(rsrec_t *h, int kk, int nncols, const int32_t *nn, const int32_t *iz, int nmax, int ntype);

/* OPTIONAL locality hint: Cartesian atom positions, lattice%cr(3,kk) (lattice.f90:138-239), any unit.  The recursion's
 * arithmetic never reads them; they only decide the ORDER in which atoms are processed so that neighbouring atoms share
 * their psi-block gathers in L2.  Without the hint an ordering is derived from graph distances.  Results do not depend on it
 * beyond summation order of the reductions. */
int rsrec_set_positions(rsrec_t *h, const double *cr);

/* Operator blocks read by the recursion: hamiltonian%ee, %lsham, %eeo, %enim, %hall, %hallo
 * (hamiltonian.f90:51-64, shapes :290-301).  Must be called again whenever the caller rebuilt them
 * (self.f90:777-797 rebuilds before every recur* call).
 *   ee, eeo    : complex (18,18,nslots,ntype)      hall, hallo : complex (18,18,nslots,nmax) (NULL if nmax = 0)
 *   lsham, enim: complex (18,18,ntype)
 *   hoh != 0 selects H = h - h*o*h + e_nu + l.s (hop_b_hoh, recursion.f90:1411); eeo/enim/hallo may be NULL otherwise.
 *   nsp: control%nsp (1..4); only the scalar recursion reads it (hop is a no-op unless nsp = 1, recursion.f90:3326). */
int rsrec_set_hamiltonian(rsrec_t *h, int nslots, int hoh, int nsp, const double *ee, const double *lsham,
                          const double *eeo, const double *enim, const double *hall, const double *hallo);

/* The raw blocks themselves, assembled on the device: what build_bulkham (hamiltonian.f90:1553-1616; part = 0: one class per atom type,
 * result ee / eeo) and build_locham (:1618-1667; part = 1: one class per impurity atom, result hall / hallo) do after chbar_nc:
 *   blocks(:,:,m,c)   = [[H0 + Hz, Hx - i Hy], [Hx + i Hy, H0 - Hz]]  with (Hx, Hy, Hz, H0) = hmag(:,:,m,1..4) of class c   (:1565-1570, :1631-1636)
 *   blocks_o(:,:,m,c) = blocks(:,:,m,c) * obarm(:,:,nbr_type(m,c)),  zero where nbr_type(m,c) = 0                          (:1599, :1654; hoh only)
 *   hmag     : complex (9,9,nslots,4,ncls) -- the scratch array chbar_nc fills, collected per class by the caller
 *   nbr_type : int32 (nslots,ncls), atom type (1..ntype) behind slot m of the class atom (slot 1: the atom itself), 0 = no atom (:1586-1597)
 *   obarm    : complex (18,18,ntype) (build_obarm, hamiltonian.f90:1486-1506);  nbr_type / obarm / blocks_o may be NULL when hoh = 0
 *   blocks, blocks_o : complex (18,18,nslots,ncls) out -- the caller's ee / eeo or hall / hallo (the reference's host routines read them too).
 * The device copies stay behind: a following rsrec_set_hamiltonian that is handed exactly these arrays (compared bit for bit) takes them
 * from there instead of uploading; arrays that were edited on the host in between are uploaded as always.  Needs no lattice. */
int rsrec_assemble_blocks(rsrec_t *h, int part, int ncls, int nslots, int hoh, const double *hmag, const int32_t *nbr_type,
                          const double *obarm, int ntype, double *blocks, double *blocks_o);

/* Block Lanczos recursion for `nsites` independent chains seeded with psi(:,:,seed) = I18.
 * Replaces recur_b (recursion.f90:1807-1866) + crecal_b (:1873-1973) + hop_b/hop_b_hoh (:1560/:1411).
 *   seed_atoms : int32 (nsites), 1-based cluster atom numbers (lattice%irec(start_atom:end_atom))
 *   a_b, b2_b  : complex (18,18,lld,nsites) out; a_b(:,:,lld,:) = 0 and b2_b(:,:,1,:) = I as in :1836-1837.
 * b2_b holds B^2 (NOT its square root: the reference takes the root later in zsqr). */
int rsrec_block_lanczos(rsrec_t *h, int nsites, const int32_t *seed_atoms, int lld, double *a_b, double *b2_b);

/* Same with general seeds: chain c starts from coef(k,c) * I18 placed on atom seed(k,c), k = 1..nseed.
 * Replaces the four-chain seeds of recur_b_ij (recursion.f90:1655-1800: (psi_i +- psi_j)/sqrt2, (psi_i +- i psi_j)/sqrt2).
 * Seeds are ASSIGNED in order, psi(l,l,seed(k,c)) = coef(k,c): a later seed on the same atom overwrites (as :1709-1714 do).
 *   seed_atoms : int32 (nseed, nchains);  seed_coef : complex (nseed, nchains) */
int rsrec_block_lanczos_seeded(rsrec_t *h, int nchains, int nseed, const int32_t *seed_atoms, const double *seed_coef,
                               int lld, double *a_b, double *b2_b);

/* recur_b with hamiltonian%local_axis = T (non-collinear runs, recursion.f90:1830-1832), every site in ONE batched call.
 * The reference rotates all blocks into the spin frame of site i's moment before that site's chain (rotate_to_local_axis,
 * hamiltonian.f90:2442-2465: ee, hall, eeo, hallo, enim -- NOT lsham) and runs the sites one after the other.  Here the
 * operator is set ONCE in the GLOBAL frame (rsrec_set_hamiltonian with ee_glob, hall_glob, eeo_glob, hallo_glob, enim_glob and
 * lsham) and every chain carries its rotation:
 *   rot : complex (18,18,nsites), R_i = ROTMAT(alpha_i, beta_i, 0) of site i's moment direction (math.f90:2026, car2sph :2155)
 * The chain of the rotated operator equals the chain of the global operator with the on-site term R_i l.s R_i^H, conjugated:
 * a_b(:,:,:,i) = R_i^H A R_i, b2_b likewise -- the library does both; outputs are in the local frame of each site like the
 * reference's.  a_b, b2_b : complex (18,18,lld,nsites) out. */
int rsrec_block_lanczos_local_axis(rsrec_t *h, int nsites, const int32_t *seed_atoms, const double *rot, int lld, double *a_b, double *b2_b);

/* The per-site result the ranks exchange after the recursion, packed ON THE DEVICE from the coefficients the last
 * rsrec_block_lanczos call left there: a(ll,l,site) = Re a_b(l,l,ll,site), b2 likewise (recursion.f90:1850-1851).
 * The reference gathers per-site arrays with MPI_ALLREDUCE(MPI_IN_PLACE, ..., MPI_SUM) on zero-padded images
 * (bands.f90:271-274); this writes such an image for this rank: sites site_offset+1 .. site_offset+nsites (the rank's
 * start_atom-1 from rsrec_site_partition) are filled, every other site is zero, so one RCCL all-reduce(sum) over the
 * ranks' images is the gather.
 *   a_img, b2_img : real (lld, 18, nsites_total) out -- HOST or DEVICE memory (detected): a device buffer (e.g. the
 *   tensor handed to the collective) is written in place, no host round trip. */
int rsrec_pack_diag(rsrec_t *h, int site_offset, int nsites_total, double *a_img, double *b2_img);

/* In-place principal square root of `nmat` Hermitian 18x18 matrices: b2_b <- sqrt(b2_b).
 * Replaces zsqr (recursion.f90:1980-2023). */
int rsrec_zsqr(rsrec_t *h, int nmat, double *b2_b);

/* Green function from the block coefficients -- the stage right after the recursion.
 * Replaces green%block_green (green.f90:588-621) + green%bgreen (:1191-1339) for the sites of this rank:
 *   g0(:,:,ie,site) = B_1^H [ E - A_1 - B_2^H [ ... ]^-1 B_2 ]^-1 B_1 , levels lld-1 .. 1, closed by the square-root terminator.
 *   ene    : real (nen) energy mesh, energy%ene(1:channels_ldos+10) (energy.f90:205-207)
 *   eta    : complex broadening added to E on the diagonal (block_green: 0; block_green_eta: the caller's eta)
 *   sym_term: control%sym_term (orbital-independent terminator, green.f90:1268-1277)
 *   a_inf, b_inf : real (18,18,nsites) from recursion%get_terminf (recursion.f90:2092) or rsrec_terminator
 *   a_b    : complex (18,18,lld,nsites) as returned by rsrec_block_lanczos
 *   b_sqrt : complex (18,18,lld,nsites) = b2_b AFTER rsrec_zsqr (self.f90:829 calls zsqr before block_green)
 *   g0     : complex (18,18,nen,nsites) out */
int rsrec_block_green(rsrec_t *h, int nsites, int lld, int nen, const double *ene, double eta_re, double eta_im, int sym_term,
                      const double *a_inf, const double *b_inf, const double *a_b, const double *b_sqrt, double *g0);

/* Band-dependent terminator coefficients of the block recursion.  Replaces recursion%get_terminf (recursion.f90:2092-2135) with
 * get_cinf (:2030-2086), bpopt (:3540-3580: Beer-Pettifor iteration) and emami (:3589-3700: extreme eigenvalues of the tridiagonal
 * chain by Sturm bisection): one GPU thread per matrix element and site, the reference's operations in the reference's order.
 *   a_b    : complex (18,18,lld,nsites);  b_sqrt : complex (18,18,lld,nsites) = b2_b after zsqr (self.f90:829)
 *   a_inf, b_inf : real (18,18,nsites) out;  a_inf0, b_inf0 : real (nsites) out (mean diagonals; may be NULL) */
int rsrec_terminator(rsrec_t *h, int nsites, int lld, const double *a_b, const double *b_sqrt, double *a_inf, double *b_inf,
                     double *a_inf0, double *b_inf0);

/* The stage behind the SCALAR recursion (control%recur = 'lanczos'): replaces dos%density (density_of_states.f90:248-363) with its
 * continued fraction bprldos (:370-404), which green%sgreen (green.f90:628-705) calls for every site and direction and turns into g0.
 * Per chain (orbital, site, direction): the Beer-Pettifor band edges (bpOPT on sqrt(b2), recursion.f90:3540; x 1.01 on orbitals 1 and 10),
 * then one scalar continued fraction per energy, closed by the square-root terminator of that band -- one GPU thread each, the reference's
 * operations in the reference's order.
 *   a, b2  : real (llmax,18,nsites,nmdir) = recursion%a, %b2 (recursion.f90:3485 recur);  lld = control%lld <= llmax
 *   ene    : real (npts) = energy%ene(1:channels_ldos+10);  dw_l, cshi : real (18,nsites) = potential%dw_l, %cshi of the sites' atoms
 *   tdens  : real (18,npts,nsites,nmdir) out = density's `tdens` for every (site, direction) */
int rsrec_scalar_density(rsrec_t *h, int nsites, int nmdir, int llmax, int lld, const double *a, const double *b2, int npts, const double *ene,
                         const double *dw_l, const double *cshi, double *tdens);

/* The whole LDOS stage for the sites of the LAST rsrec_block_lanczos call, from the coefficients that call left on the device
 * (nothing is uploaded but the energy mesh): zsqr (recursion.f90:1980) -> get_terminf (:2092) -> green%bgreen (green.f90:1191,
 * eta / sym_term as in rsrec_block_green) -> the reduction of bands%calculate_fermi (bands.f90:258-268),
 *   dosial(ia,j,i) = -Im g0(j,j,i,ia)/pi,  dosia(ia,i) = sum_j dosial,  dtot(i) = sum_ia dosia   (summed in the reference's order).
 * Outputs are the zero-padded images the reference all-reduces over the ranks (bands.f90:271-274): this rank's sites are
 * site_offset+1 .. site_offset+nsites of nsites_total, every other site is zero.
 *   dtot : real (nen);  dosia : real (nsites_total, nen);  dosial : real (nsites_total, 18, nen)  -- HOST or DEVICE memory (all three
 *   alike, detected): device buffers are written in place and can be handed to the collective without a host round trip.
 *   a_inf, b_inf : real (18,18,nsites) host out, the terminators used (may be NULL).
 * 18 doubles per site and energy leave the GPU instead of the 648 of g0. */
int rsrec_block_ldos(rsrec_t *h, int nen, const double *ene, double eta_re, double eta_im, int sym_term, int site_offset, int nsites_total,
                     double *dtot, double *dosia, double *dosial, double *a_inf, double *b_inf);

/* Green function from the Chebyshev moments.  Replaces green%chebyshev_green (green.f90:1030-1108) for the sites of this rank:
 *   g0(:,:,ie,site) = sum_i mu_n(:,:,i,site) k_i (-i) exp(-i (i-1) acos w_ie) / sqrt(a^2 - (e_ie - b)^2),  w = (e - b)/a,
 *   k = Jackson kernel (math.f90:1641) times 2 for i > 1; a, b from energy_min/max as in rsrec_chebyshev.
 *   mu_n : complex (18,18,2*lld+2,nsites) as returned by rsrec_chebyshev;  g0 : complex (18,18,nen,nsites) out */
int rsrec_chebyshev_green(rsrec_t *h, int nsites, int lld, int nen, const double *ene, double energy_min, double energy_max,
                          const double *mu_n, double *g0);

/* Chebyshev (KPM, moment doubling) recursion.  Replaces chebyshev_recur (recursion.f90:3057-3130) with
 * cheb_0th_mom (:2145), cheb_1st_mom[_hoh] (:2169/:2245), chebyshev_recur_ll[_hoh] (:2495/:2605).
 *   a, b : scale and shift, a = (energy_max-energy_min)/(2-0.3), b = (energy_max+energy_min)/2 (:3078-3079)
 *   mu_n : complex (18,18,2*lld+2,nsites) out
 * Returns RSREC_ERR_DIVERGED if sum(real(mu_n(:,:,2ll+2))) > 1000 at any step (:2594). */
int rsrec_chebyshev(rsrec_t *h, int nsites, const int32_t *seed_atoms, int lld, double a, double b, double *mu_n);

/* Same with general seeds, psi0(l,l,seed(k,c)) = coef(k,c) in order.  Replaces chebyshev_recur_ij (recursion.f90:2376-2487),
 * whose four chains per pair start from (psi_i +- psi_j)/sqrt2 and (psi_i +- i psi_j)/sqrt2; both new moments are tested
 * against the 1000 bound (:2484).  seed_atoms int32 (nseed, nchains), seed_coef complex (nseed, nchains), nseed <= 8. */
int rsrec_chebyshev_seeded(rsrec_t *h, int nchains, int nseed, const int32_t *seed_atoms, const double *seed_coef,
                           int lld, double a, double b, double *mu_n);

/* Stochastic Kubo-Bastin double moments.  Replaces compute_moments_stochastic (recursion.f90:979-1234) together with the
 * whole-vector products it is built from: ham_vec_matmul (:913), ham_hoh_vec_matmul (:785), velo_vec_matmul (:587),
 * velo_hoh_vec_matmul (:656):
 *   mu(:,:,n,m,i) = sum_k [ T_{m-1}(H~) r_i ]_k^H  [ v_a T_{n-1}(H~) v_b r_i ]_k ,   H~ = (H - b)/a ,  n, m = 1..cond_ll
 * with H the operator set by rsrec_set_hamiltonian (hoh included), T the Chebyshev polynomials (three-term recurrence on whole
 * vectors) and the sum running over every atom k of the lattice.
 *   nvec vectors; vector i starts from psiref(l,l,seed(k,i)) = coef(k,i), k = 1..nseed (seed = 0: entry unused):
 *     cond_calctype = 'per_type'   : one seed, the type's atom lattice%atlist(i), coefficient 1            (:1093-1102)
 *     cond_calctype = 'random_vec' : nseed = kk, coefficient exp(2 pi i rng_k) / sqrt(kk) from the caller's generator (:1103-1114)
 *   seed_atoms : int32 (nseed, nvec);  seed_coef : complex (nseed, nvec)
 *   a, b : scale and shift as in rsrec_chebyshev (:1023-1024)
 *   v_a, v_b   : complex (18,18,nslots,ntype), hamiltonian%v_a / %v_b as setup_kubo_operators (:242) left them
 *   vo_a, vo_b : complex (18,18,nslots,ntype), hamiltonian%vo_a / %vo_b (hoh only, else NULL)
 *   mu_nm : complex (18,18,cond_ll,cond_ll,nvec) out = recursion%mu_nm_stochastic
 * The velocity operators act on the bulk (per-type) atoms only, as in the reference (:591, "not yet implemented" for the
 * per-atom region): rows of v psi on the first nmax atoms are zero. */
int rsrec_kubo_moments(rsrec_t *h, int nvec, int nseed, const int32_t *seed_atoms, const double *seed_coef, int cond_ll, double a, double b,
                       const double *v_a, const double *vo_a, const double *v_b, const double *vo_b, double *mu_nm);

/* One whole-vector product on caller arrays psi(18,18,kk) (complex, the reference's layout):
 *   vel = 0 : psi_out = (H psi_in - b psi_in)/a      ham_vec_matmul (:913) / ham_hoh_vec_matmul (:785); v_op, vo_op ignored
 *   vel = 1 : psi_out = V psi_in                      velo_vec_matmul (:587, 'n') / velo_hoh_vec_matmul (:656) with v_op (and vo_op with hoh)
 * (The type-bound procedures of those names can be overridden with this; chebyshev_orbital_mod :2834 then runs its products on the GPU.) */
int rsrec_apply_operator(rsrec_t *h, int vel, const double *v_op, const double *vo_op, const double *psi_in, double *psi_out, double a, double b);

/* Scalar Haydock recursion, one chain per (site, orbital).  Replaces recur (recursion.f90:3485-3532),
 * crecal (:3423-3478), hop (:3310-3416).
 *   a, b2 : real (llmax,18,nsites) out (the (:,:,:,1) plane of the reference's a/b2); rows > lld are zeroed. */
int rsrec_scalar_lanczos(rsrec_t *h, int nsites, const int32_t *seed_atoms, int lld, int llmax, double *a, double *b2);

/* Site partition of get_mpi_variables (mpi.f90:32-58): 1-based inclusive range owned by `rank`. */
void rsrec_site_partition(int rank, int nprocs, int nsites, int *start_atom, int *end_atom);

/* Last error text of this handle (NUL-terminated, truncated to n). */
int rsrec_last_error(rsrec_t *h, char *buf, size_t n);

/* chebyshev_orbital_mod (recursion.f90:2834-3049; called at calculation.f90:1256), the moment part (:2893-3013), with the seeds advanced
 * together as chains and every vector resident on the device.  For seed atom s:  psiref = 1 on s;
 * left = i (Y H~ X - X H~ Y) psiref  with X, Y = alat cr(1,:), alat cr(2,:) and H~ = ham_vec_matmul (the plain operator also when hoh is
 * set);  v_1 = psiref, v_2 = H~' v_1, v_n = 2 H~' v_{n-1} - v_{n-2}  (H~' = ham_hoh_vec_matmul with hoh);  mu(:,:,n) = sum_k left_k^H v_n,k.
 *   cr      : lattice%cr(3,kk), units of alat;   a, b : scale and shift of H~ = (H - b)/a  (:2869-2870)
 *   mu_orb  : complex (18,18,lld): the SUM over the seeds of the call in seed order (the reference loops over all kk atoms and divides
 *             by kk afterwards, :3006; its own accumulator is never zeroed, :2907 -- here the sum starts from zero)
 *   mu_seed : optional complex (18,18,lld,nseeds): the contribution of every seed */
int rsrec_orbital_moments(rsrec_t *h, int nseeds, const int32_t *seed_atoms, int lld, double a, double b, const double *cr, double alat,
                          double *mu_orb, double *mu_seed);

/* The Chebyshev counterpart of rsrec_pack_diag: the moments mu_n(18,18,2 lld + 2,site) of the last rsrec_chebyshev call, as they lie on
 * the device, inside a zero image over all sites (shape (18,18,2 lld + 2,nsites_total) complex; device or host memory) -- the buffer a
 * sum all-reduce turns into the all-gather of recursion.f90:1790-1793 (commented-out MPI_Allgather of the coefficients). */
int rsrec_pack_moments(rsrec_t *h, int site_offset, int nsites_total, double *mu_img);

/* Library-level communicator: the one exchange of the path -- MPI_ALLREDUCE(MPI_IN_PLACE, ..., MPI_SUM) on zero-padded per-site arrays
 * (bands.f90:271-274; sites are dealt to the ranks by mpi.f90:32-58 = rsrec_site_partition) -- as ONE RCCL all-reduce over xGMI,
 * without MPI or torch on the host.  RCCL is bound with dlopen on first use.  One rank per GPU; the handle's device is the rank's GPU.
 *   rsrec_comm_unique_id : one rank creates the id (RSREC_COMM_ID_BYTES bytes) and hands it to the others (MPI_Bcast, a file, ...)
 *   rsrec_comm_init      : collective; same id on every rank
 *   rsrec_comm_init_file : the same with the id exchanged through `path` (rank 0 writes it, the others wait up to timeout_s seconds)
 *   rsrec_allreduce_sum  : in-place sum of n doubles over the ranks; buf = device memory (the images of rsrec_pack_diag /
 *                          rsrec_pack_moments / rsrec_block_ldos, reduced where they lie) or host memory (staged).  Identity without a
 *                          communicator, like the reference built without MPI.
 *   rsrec_comm_size      : rank and number of ranks of the handle's communicator (0, 1 without one) */
#define RSREC_COMM_ID_BYTES 128
int rsrec_comm_unique_id(char *id);
int rsrec_comm_init(rsrec_t *h, int rank, int nranks, const char *id);
int rsrec_comm_init_file(rsrec_t *h, int rank, int nranks, const char *path, double timeout_s);
int rsrec_allreduce_sum(rsrec_t *h, double *buf, size_t n);
int rsrec_comm_size(rsrec_t *h, int *rank, int *nranks);
int rsrec_comm_destroy(rsrec_t *h);

/* ---- tuning / measurement (not part of the reference interface) ---- */
/* key/value knobs (defaults in brackets; everything but "batch" and "kernels" exists for A/B measurements and tests):
 *   "batch"      chains advanced together per launch [0 = auto: up to 64, bounded by free device memory]
 *   "kernels"    0 = auto, 1 = FP64 VALU kernel set (the reference's layout and operation order; any stencil), 2 = matrix-core set [0]
 *   "spmm5"      SpMM of the matrix-core set: 0 = small-launch kernel k_spmm4<4> (LayoutRM vectors) whenever it applies, 1 = by launch size
 *                (k_spmm5 on CI vectors from 4096 groups per launch; always for hoh and local-axis runs), 2 = always k_spmm5 [2]
 *   "side_stream" reduction + eigen-solve of B_{n+1} on a second HIP stream, concurrent with the next H|u>; the chain-octet launch beside
 *                the main H|psi> launch [1]
 *   "graph"      the level loop of a block-Lanczos call as one HIP graph, replayed while lattice, seeds, depth, buffers and options stay
 *                the same: 0 = never, 1 = calls of up to 8 chains, 2 = every single-batch call [1]
 *   "nblk"       workgroups per chain / 2 of the reduction-bearing kernels [0 = by batch size]
 *   "orth3"      k_mfma_orth3: 1 = one 512-register wave per SIMD (tables in registers), 2 = two waves per SIMD (tables in LDS) [1]
 *   "cheb_fused" Chebyshev step inside the SpMM epilogue (H psi never written): 0 / 1 [1]
 *   "chain_fold" chains per k_spmm5 workgroup [1],  "s5_cap" cap on k_spmm5 workgroups per chain [0 = none],
 *   "s5_lds"     k_spmm5 with the operator fragments in LDS: 0 = never, 1 = operators with one class of atoms, whenever their stream fits,
 *                2 = also operators with several classes (one run of groups per class of the class-sorted atom list; slower) [1]
 *   "s5_queue"   the LDS form as persistent workgroups (one per CU) with per-(chain, XCD) group counters: 0 = never, 1 = launches of >= 256
 *                workgroups, 2 = always [1];  "s5_waves" waves per persistent workgroup, 8 or 4 [8];  "s5_split" 3 = a wave of the persistent form takes a
 *                third of a group's nine tiles (bitwise the same results, measured slower: DESIGN.md; s5_waves 8 / 12 then) [0];  "s5_run_min" smallest class run
 *                (groups) that gets LDS workgroups of its own under s5_lds = 2 [0 = by launch size]
 *   "s5_spin_xcd" persistent form on collinear operators: 1 = even XCDs serve output spin 0 and odd XCDs spin 1 (an XCD's L2 then holds one
 *                spin half of the neighbour blocks; the round-2 default), 0 = both spins on every XCD (2-4 % faster, round 3) [0]
 *   "s5_octet"   atoms with their own operator blocks (nmax) from which their groups are formed over 8 CHAINS of the batch (they then share
 *                the atom's operator fragments the way 8 atoms of a type do) once every chain's region covers the lattice; 0 = never [8]
 *   "s5_host_emit" 1 = swizzle k_spmm5's operator streams on the host instead of assembling them on the device (cross-check) [0]
 *   "orth_oop"   1 = the orthogonalisation pass writes u_{n+1} into a third u vector instead of over u_{n-1} (faster on the HBM, one more work vector) [1]
 *   "sat_pct"    a chain whose region holds at least this share (per cent) of the lattice runs on the list of ALL atoms instead of its own [100]
 *   "kubo_lchunk" rsrec_kubo_moments: left vectors held on the device at a time [0 = as many as fit];  "kubo_vbatch" vectors of a call advanced
 *                together as the chains of every launch [0 = up to 8, as many as fit beside a whole left matrix each] */
int rsrec_set_option(rsrec_t *h, const char *key, long value);
/* Timing of the last recursion call, measured with HIP events on the engine's own stream:
 *   out[0] total device ms, out[1] ms in the H|psi> kernels, out[2] number of H|psi> launches,
 *   out[3] atom-steps processed (sum over chains and steps of active atoms), out[4] block multiplies in H|psi>,
 *   out[5] ms in the remaining recursion kernels, out[6] host ms (region bookkeeping + transfers),
 *   out[7] 1 if the timed H|psi> kernel also forms the A_n partial (VALU / fused variants), else 0,
 *   out[8] matrix flops EXECUTED by the timed k_spmm5 launches (padding of the MFMA tiles included, structural zeros of spin-diagonal
 *          blocks not: they are skipped), per operator class of the groups; 0 for the other kernels.
 *   out[9] flops of the H|psi> applications that the operator's BLOCK STRUCTURE requires: out[4] counts the reference's zgemm on full
 *          18x18 blocks (46 656 flop each, recursion.f90:1618); a spin-diagonal block (every hopping block of a collinear magnet,
 *          hamiltonian.f90:1553-1617) needs 23 328, a spin-mixing block 46 656 -- the unit roofline fractions are quoted in.
 *   out[10] block arrays (0..4: ee, eeo, hall, hallo) the last rsrec_set_hamiltonian took from rsrec_assemble_blocks' device copies.
 *   out[11] H|psi> launches of the last call in which the atoms with their own operator blocks were grouped over 8 chains (option "s5_octet").
 * After rsrec_block_green: out[0] = kernel + transfers, out[1] = the Green kernel alone. */
int rsrec_get_timing(rsrec_t *h, double *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* RSREC_H */
