/*
 * pp_oracle.c - CPU restatement of the reference's PP / P3M short-range force.
 *
 * TEST INFRASTRUCTURE ONLY (see pp_oracle.h).  Build with -ffp-contract=off so that
 * every multiply and add below rounds separately, the way the statements are written
 * in the reference.
 *
 * Citations are relative to the reference tree.
 */
#include "pp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* src/math_utils.h:11-16: one image only, strict compares, +-L/2 left as it is. */
double orc_periodic_displacement(double dx, double domain_size) {
    const double half = 0.5 * domain_size;
    if (dx > half) dx = dx - domain_size;
    if (dx < -half) dx = dx + domain_size;
    return dx;
}

/* src/config.cpp:73-99, statement by statement. */
void orc_config_derive(const orc_config_in* in, orc_config_out* o) {
    o->cell_size = in->domain_size / in->mesh_size;                       /* :74 */
    o->cell_volume = o->cell_size * o->cell_size * o->cell_size;          /* :75 */
    o->omega_dm = in->omega_m - in->omega_baryon;                         /* :76 */
    o->cutoff_radius = in->cutoff_radius_cells * o->cell_size;            /* :77 */
    o->cutoff_radius_squared = o->cutoff_radius * o->cutoff_radius;       /* :78 */
    o->cutoff_transition_width =
        in->cutoff_transition_width_factor * o->cutoff_radius;            /* :79 */
    o->r_switch_start = o->cutoff_radius - o->cutoff_transition_width;    /* :80 */
    o->r_switch_start_sq = o->r_switch_start * o->r_switch_start;         /* :81 */
    o->num_dm_particles = in->n_per_side * in->n_per_side * in->n_per_side; /* :83 */
    o->pad = 0;
    const double total_mass = 1.0;                                        /* :85 */
    o->G = pow(in->domain_size, 3) / (6.0 * M_PI * total_mass);           /* :86 */
    const double baryon_fraction = in->omega_baryon / in->omega_m;        /* :88 */
    const double dm_total_mass = total_mass * (1.0 - baryon_fraction);    /* :89 */
    o->dm_particle_mass = dm_total_mass / o->num_dm_particles;            /* :90 */
    o->gas_total_mass = total_mass * baryon_fraction;                     /* :91 */
    const double spacing =
        in->domain_size / cbrt((double)o->num_dm_particles);              /* :93-94 */
    o->softening_squared = pow(spacing / 50.0, 2);                        /* :96 */
    const double dynamical_time = 1.0 / sqrt(o->G);                       /* :97 */
    o->fixed_dt = in->dt_factor * dynamical_time;                         /* :98 */
}

/* The hoisted scalars of src/particles.cpp:143-152. */
void orc_params_from_config(const orc_config_in* in, int use_pm, orc_params* p) {
    orc_config_out d;
    orc_config_derive(in, &d);
    p->domain_size = in->domain_size;
    p->G = d.G;
    p->soft_sq = d.softening_squared;
    p->pm_soft_sq = (0.5 * d.cell_size) * (0.5 * d.cell_size);            /* :146-147 */
    p->r_switch = d.r_switch_start;
    p->r_switch_sq = d.r_switch_start_sq;
    p->cutoff_sq = d.cutoff_radius_squared;
    p->trans_width = d.cutoff_transition_width;
    p->cell_size = d.cell_size;
    p->mesh_size = in->mesh_size;
    p->use_pm = use_pm ? 1 : 0;
}

/*
 * One source's contribution to one target: src/particles.cpp:321-359 (CPU branch),
 * identical arithmetic at :212-254 (offload branch).  Returns 1 when the pair passed
 * the cutoff test.
 */
static inline int pair_term(const orc_params* p, double xi, double yi, double zi,
                            double xj, double yj, double zj, double mj, double* sx,
                            double* sy, double* sz) {
    const double L = p->domain_size;
    const double dx = orc_periodic_displacement(xj - xi, L);              /* :321 */
    const double dy = orc_periodic_displacement(yj - yi, L);              /* :322 */
    const double dz = orc_periodic_displacement(zj - zi, L);              /* :323 */
    const double dist_sq = dx * dx + dy * dy + dz * dz;                   /* :324 */

    if (p->use_pm && dist_sq > p->cutoff_sq) return 0;                    /* :326-328 */

    double S = 1.0;                                                       /* :330 */
    if (p->use_pm && dist_sq > p->r_switch_sq) {                          /* :331 */
        const double dist = sqrt(dist_sq);                                /* :332 */
        const double t = (dist - p->r_switch) / p->trans_width;           /* :333 */
        S = 2.0 * (t * t * t) - 3.0 * (t * t) + 1.0;                      /* :334 */
    }

    double pmx = 0.0, pmy = 0.0, pmz = 0.0;                               /* :337 */
    if (p->use_pm) {                                                      /* :339 */
        const double q = dist_sq + p->pm_soft_sq;                         /* :341 */
        const double rq = sqrt(q);                                        /* :342 */
        const double a = p->G * mj / q;                                   /* :343 */
        pmx = a * dx / rq;                                                /* :344 */
        pmy = a * dy / rq;                                                /* :345 */
        pmz = a * dz / rq;                                                /* :346 */
    }

    const double q = dist_sq + p->soft_sq;                                /* :349 */
    const double rq = sqrt(q);                                            /* :350 */
    const double a = p->G * mj / q;                                       /* :351 */
    const double ppx = a * dx / rq;                                       /* :353 */
    const double ppy = a * dy / rq;                                       /* :354 */
    const double ppz = a * dz / rq;                                       /* :355 */

    *sx += S * (ppx - pmx);                                               /* :357 */
    *sy += S * (ppy - pmy);                                               /* :358 */
    *sz += S * (ppz - pmz);                                               /* :359 */
    return 1;
}

/* src/particles.cpp:134-137 */
static int search_radius(const orc_params* p) {
    return p->use_pm ? (int)ceil(sqrt(p->cutoff_sq) / p->cell_size) : p->mesh_size / 2;
}

/*
 * Head/next linked list exactly as src/particles.cpp:50,60-62,81-83,94-96 builds it:
 * particles scanned in ascending index and pushed at the front, so each cell's chain is
 * in descending particle index; cell id = iz*N*N + iy*N + ix with x fastest.
 */
static int build_cell_list(const orc_params* p, size_t n, const double* x, const double* y,
                           const double* z, int** head_out, int** next_out) {
    const int N = p->mesh_size;
    const size_t ncell = (size_t)N * N * N;
    int* head = (int*)malloc(ncell * sizeof(int));
    int* next = (int*)malloc((n ? n : 1) * sizeof(int));
    if (!head || !next) { free(head); free(next); return -1; }
    for (size_t c = 0; c < ncell; ++c) head[c] = -1;
    for (size_t i = 0; i < n; ++i) {
        const int ix = (int)(x[i] / p->cell_size);                        /* :60 */
        const int iy = (int)(y[i] / p->cell_size);                        /* :61 */
        const int iz = (int)(z[i] / p->cell_size);                        /* :62 */
        const int ix0 = (ix + N) % N, iy0 = (iy + N) % N, iz0 = (iz + N) % N; /* :81-83 */
        const int cell = iz0 * N * N + iy0 * N + ix0;                     /* :94 */
        next[i] = head[cell];                                             /* :95 */
        head[cell] = (int)i;                                              /* :96 */
    }
    *head_out = head;
    *next_out = next;
    return 0;
}

#define ORC_MAX_NEIGHBORS 1500 /* src/particles.cpp:280 */

int orc_pp_accumulate(const orc_params* p, size_t n, const double* x, const double* y,
                      const double* z, const double* m, size_t tgt_begin, size_t n_tgt,
                      double* ax, double* ay, double* az, int traversal, int nthreads) {
    if (!p || tgt_begin + n_tgt > n) return -1;
    if (n_tgt == 0) return 0;
    if (!x || !y || !z || !m || !ax || !ay || !az) return -1;
    if (traversal == ORC_EXACT_CELLS && !p->use_pm) traversal = ORC_EXACT;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif

    if (traversal == ORC_EXACT) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
        for (size_t t = 0; t < n_tgt; ++t) {
            const size_t i = tgt_begin + t;
            const double xi = x[i], yi = y[i], zi = z[i];
            double sx = 0.0, sy = 0.0, sz = 0.0;
            for (size_t j = 0; j < n; ++j) {
                if (j == i) continue;                                     /* :202,:302 */
                pair_term(p, xi, yi, zi, x[j], y[j], z[j], m[j], &sx, &sy, &sz);
            }
            ax[t] += sx;                                                  /* :362-364 */
            ay[t] += sy;
            az[t] += sz;
        }
        return 0;
    }

    if (p->mesh_size <= 0 || !(p->cell_size > 0.0)) return -1;
    int *head = NULL, *next = NULL;
    if (build_cell_list(p, n, x, y, z, &head, &next)) return -1;
    const int N = p->mesh_size;
    const int r = search_radius(p);

#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
    for (size_t t = 0; t < n_tgt; ++t) {
        const size_t i = tgt_begin + t;
        const double xi = x[i], yi = y[i], zi = z[i];
        const int ix = (int)(xi / p->cell_size);                          /* :276-278 */
        const int iy = (int)(yi / p->cell_size);
        const int iz = (int)(zi / p->cell_size);
        double sx = 0.0, sy = 0.0, sz = 0.0;
        int gathered = 0;

        if (traversal == ORC_EXACT_CELLS) {
            /* Distinct cells only: offsets clipped so that no cell repeats. */
            const int span = (2 * r + 1 >= N) ? N : 2 * r + 1;
            const int lo = (span == N) ? 0 : -r;
            for (int a = 0; a < span; ++a)
                for (int b = 0; b < span; ++b)
                    for (int c = 0; c < span; ++c) {
                        const int cx = (((ix + lo + a) % N) + N) % N;
                        const int cy = (((iy + lo + b) % N) + N) % N;
                        const int cz = (((iz + lo + c) % N) + N) % N;
                        for (int j = head[cz * N * N + cy * N + cx]; j != -1; j = next[j]) {
                            if ((size_t)j == i) continue;
                            pair_term(p, xi, yi, zi, x[j], y[j], z[j], m[j], &sx, &sy, &sz);
                        }
                    }
        } else {
            /* :287-312 (CPU) / :187-259 (offload): x offset outermost, z innermost. */
            for (int dxc = -r; dxc <= r; ++dxc)
                for (int dyc = -r; dyc <= r; ++dyc)
                    for (int dzc = -r; dzc <= r; ++dzc) {
                        const int cx = (ix + dxc + N) % N;                /* :293 */
                        const int cy = (iy + dyc + N) % N;                /* :294 */
                        const int cz = (iz + dzc + N) % N;                /* :295 */
                        const int cell = cz * N * N + cy * N + cx;        /* :296-297 */
                        for (int j = head[cell]; j != -1; j = next[j]) {
                            if ((size_t)j == i) continue;
                            if (traversal == ORC_REF_CPU) {
                                if (gathered >= ORC_MAX_NEIGHBORS) continue; /* :302 */
                                ++gathered;
                            }
                            pair_term(p, xi, yi, zi, x[j], y[j], z[j], m[j], &sx, &sy, &sz);
                        }
                    }
        }
        ax[t] += sx;
        ay[t] += sy;
        az[t] += sz;
    }
    free(head);
    free(next);
    return 0;
}

/* Independent check: same physics, long double, textbook form G m d / q^(3/2). */
int orc_pp_accumulate_ld(const orc_params* p, size_t n, const double* x, const double* y,
                         const double* z, const double* m, size_t tgt_begin, size_t n_tgt,
                         double* ax, double* ay, double* az, int nthreads) {
    if (!p || tgt_begin + n_tgt > n) return -1;
    if (n_tgt == 0) return 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    const long double L = p->domain_size, half = 0.5L * L;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
    for (size_t t = 0; t < n_tgt; ++t) {
        const size_t i = tgt_begin + t;
        long double sx = 0, sy = 0, sz = 0;
        for (size_t j = 0; j < n; ++j) {
            if (j == i) continue;
            /* the wrap decision is taken on the double difference, as the reference does */
            double d[3] = {x[j] - x[i], y[j] - y[i], z[j] - z[i]};
            long double e[3] = {(long double)x[j] - x[i], (long double)y[j] - y[i],
                                (long double)z[j] - z[i]};
            for (int c = 0; c < 3; ++c) {
                if (d[c] > (double)half) e[c] -= L;
                else if (d[c] < -(double)half) e[c] += L;
            }
            const long double r2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
            long double f;
            if (p->use_pm) {
                if ((double)r2 > p->cutoff_sq) continue;
                long double S = 1;
                if ((double)r2 > p->r_switch_sq) {
                    const long double u = (sqrtl(r2) - p->r_switch) / p->trans_width;
                    S = 2 * u * u * u - 3 * u * u + 1;
                }
                const long double q1 = r2 + p->soft_sq, q2 = r2 + p->pm_soft_sq;
                f = S * (1 / (q1 * sqrtl(q1)) - 1 / (q2 * sqrtl(q2)));
            } else {
                const long double q1 = r2 + p->soft_sq;
                f = 1 / (q1 * sqrtl(q1));
            }
            f *= (long double)p->G * m[j];
            sx += f * e[0];
            sy += f * e[1];
            sz += f * e[2];
        }
        ax[t] += (double)sx;
        ay[t] += (double)sy;
        az[t] += (double)sz;
    }
    return 0;
}

uint64_t orc_count_within_cutoff(const orc_params* p, size_t n, const double* x,
                                 const double* y, const double* z, size_t tgt_begin,
                                 size_t n_tgt, int nthreads) {
    if (!p || tgt_begin + n_tgt > n || n_tgt == 0) return 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    int *head = NULL, *next = NULL;
    if (build_cell_list(p, n, x, y, z, &head, &next)) return 0;
    const int N = p->mesh_size;
    const int r = (int)ceil(sqrt(p->cutoff_sq) / p->cell_size);
    const int span = (2 * r + 1 >= N) ? N : 2 * r + 1;
    const int lo = (span == N) ? 0 : -r;
    uint64_t total = 0;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads) reduction(+ : total)
    for (size_t t = 0; t < n_tgt; ++t) {
        const size_t i = tgt_begin + t;
        const int ix = (int)(x[i] / p->cell_size), iy = (int)(y[i] / p->cell_size),
                  iz = (int)(z[i] / p->cell_size);
        for (int a = 0; a < span; ++a)
            for (int b = 0; b < span; ++b)
                for (int c = 0; c < span; ++c) {
                    const int cx = (((ix + lo + a) % N) + N) % N;
                    const int cy = (((iy + lo + b) % N) + N) % N;
                    const int cz = (((iz + lo + c) % N) + N) % N;
                    for (int j = head[cz * N * N + cy * N + cx]; j != -1; j = next[j]) {
                        if ((size_t)j == i) continue;
                        const double dx = orc_periodic_displacement(x[j] - x[i], p->domain_size);
                        const double dy = orc_periodic_displacement(y[j] - y[i], p->domain_size);
                        const double dz = orc_periodic_displacement(z[j] - z[i], p->domain_size);
                        if (!(dx * dx + dy * dy + dz * dz > p->cutoff_sq)) ++total;
                    }
                }
    }
    free(head);
    free(next);
    return total;
}

/* src/integrator.cpp:168-181 */
void orc_kick(size_t n, double* vx, double* vy, double* vz, const double* ax,
              const double* ay, const double* az, double dt, double a, double H) {
    const double a3 = a * a * a;                                          /* :169 */
    const double dt_over_a3 = dt / a3;                                    /* :170 */
    const double dt_2H = dt * 2.0 * H;                                    /* :171 */
    for (size_t i = 0; i < n; ++i) {
        vx[i] += ax[i] * dt_over_a3 - vx[i] * dt_2H;                      /* :177 */
        vy[i] += ay[i] * dt_over_a3 - vy[i] * dt_2H;                      /* :178 */
        vz[i] += az[i] * dt_over_a3 - vz[i] * dt_2H;                      /* :179 */
    }
}

/* src/integrator.cpp:183-198 */
void orc_drift(size_t n, double* x, double* y, double* z, const double* vx,
               const double* vy, const double* vz, double dt, double domain_size) {
    const double inv_domain = 1.0 / domain_size;                          /* :184 */
    for (size_t i = 0; i < n; ++i) {
        const double nx = x[i] + vx[i] * dt;                              /* :189 */
        const double ny = y[i] + vy[i] * dt;
        const double nz = z[i] + vz[i] * dt;
        x[i] = nx - domain_size * floor(nx * inv_domain);                 /* :194 */
        y[i] = ny - domain_size * floor(ny * inv_domain);
        z[i] = nz - domain_size * floor(nz * inv_domain);
    }
}

/* src/integrator.cpp:200-213 */
double orc_max_accel_sq(size_t n, const double* ax, const double* ay, const double* az) {
    double local_max = 1e-9;                                              /* :201 */
    for (size_t i = 0; i < n; ++i) {
        const double s = ax[i] * ax[i] + ay[i] * ay[i] + az[i] * az[i];   /* :208 */
        local_max = (s > local_max) ? s : local_max;                      /* :209 */
    }
    return local_max;
}

/* src/particles.cpp:386-391, as written */
double orc_gravity_timestep(size_t n, double softening_squared, double max_accel_sq,
                            double gravity_dt_factor) {
    if (n == 0) return INFINITY;                                          /* :387 */
    const double dt_grav = sqrt(softening_squared) / sqrt(max_accel_sq);  /* :389 */
    return dt_grav * gravity_dt_factor;                                   /* :390 */
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/*
 * Standard-normal deviates exactly as libstdc++ produces them for
 *     std::default_random_engine generator(seed);            (= std::minstd_rand0)
 *     std::normal_distribution<double> distribution(0.0, 1.0);
 * which is how the reference fills its white-noise field (src/ics.cpp:62-66).  libstdc++'s
 * algorithm: Marsaglia polar method on generate_canonical<double,53>, which for minstd_rand0
 * (range 2147483646 < 2^31) consumes two engine outputs per canonical deviate; of each
 * accepted pair the second coordinate is returned first and the first is saved.
 */
static inline uint32_t minstd0_next(uint32_t* s) {
    *s = (uint32_t)(((uint64_t)(*s) * 16807u) % 2147483647u);
    return *s;
}

static inline double minstd0_canonical(uint32_t* s) {
    const double r = 2147483646.0; /* max - min + 1 */
    double sum = (double)(minstd0_next(s) - 1u);
    sum += (double)(minstd0_next(s) - 1u) * r;
    double ret = sum / (r * r);
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

void orc_libstdcxx_normal_field(uint32_t seed, size_t n, double* out) {
    uint32_t s = seed % 2147483647u;
    if (s == 0) s = 1; /* linear_congruential_engine::seed with c == 0 */
    int have_saved = 0;
    double saved = 0.0;
    for (size_t i = 0; i < n; ++i) {
        if (have_saved) {
            have_saved = 0;
            out[i] = saved;
            continue;
        }
        double x, y, r2;
        do {
            x = 2.0 * minstd0_canonical(&s) - 1.0;
            y = 2.0 * minstd0_canonical(&s) - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = sqrt(-2 * log(r2) / r2);
        saved = x * mult;
        have_saved = 1;
        out[i] = y * mult;
    }
}

/* sigma_8 normalisation integral of src/ics.cpp:38-51: the same loop, the same running k. */
static double bbks_unnormalized_pk(double k_h, double omega_m, double hubble, double n_s) {
    const double Gamma = omega_m * hubble;                                /* src/ics.cpp:14 */
    const double q = k_h / Gamma;                                         /* :17 */
    const double T_k = log(1.0 + 2.34 * q) / (2.34 * q) *
                       pow(1.0 + 3.89 * q + pow(16.1 * q, 2) + pow(5.46 * q, 3) + pow(6.71 * q, 4),
                           -0.25);                                        /* :20-23 */
    return pow(k_h, n_s) * T_k * T_k;                                     /* :26 */
}

double orc_bbks_pk(double k_h, double omega_m, double hubble, double n_s) {
    return bbks_unnormalized_pk(k_h, omega_m, hubble, n_s);
}

double orc_sigma8_amplitude(double sigma_8, double omega_m, double hubble, double n_s) {
    const double R = 8.0, dk = 0.001;                                     /* :38,:40 */
    double unnorm_variance = 0.0;
    for (double k = 0.001; k < 100.0; k += dk) {                          /* :43 */
        const double pk = bbks_unnormalized_pk(k, omega_m, hubble, n_s);
        const double kR = k * R;
        const double w = (kR < 1e-4) ? 1.0 - (kR * kR / 10.0)
                                     : 3.0 * (sin(kR) - kR * cos(kR)) / (kR * kR * kR); /* :32-33 */
        unnorm_variance += pk * w * w * k * k * dk;                       /* :46 */
    }
    unnorm_variance /= (2.0 * M_PI * M_PI);                               /* :48 */
    return (sigma_8 * sigma_8) / unnorm_variance;                         /* :51 */
}

void orc_bbks_pk_array(size_t n, const double* k_h, double omega_m, double hubble, double n_s,
                       double* out) {
    for (size_t i = 0; i < n; ++i)
        out[i] = (k_h[i] > 0.0) ? bbks_unnormalized_pk(k_h[i], omega_m, hubble, n_s) : 0.0;
}
