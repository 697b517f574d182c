/* TEST INFRASTRUCTURE — body of the CPU restatement, compiled once per precision.
 *
 * Included by rt_oracle.c with
 *     R         the arithmetic type (double | float)
 *     S(name)   name##_f64 | name##_f32
 *     M_SQRT/M_FABS/M_FMIN/M_FMAX/M_TAN/M_POW/M_SIN/M_ACOS/M_ATAN2  libm in that type
 * Expressions keep the reference's association and operand order so that the double
 * build reproduces main.cc bit for bit (no fused multiply-add: see the Makefile).
 */

typedef struct { R e[3]; } S(v3);

static inline S(v3) S(mk)(R x, R y, R z) { S(v3) r = {{x, y, z}}; return r; }
static inline S(v3) S(add)(S(v3) a, S(v3) b) { return S(mk)(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
static inline S(v3) S(sub)(S(v3) a, S(v3) b) { return S(mk)(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
static inline S(v3) S(mulv)(S(v3) a, S(v3) b) { return S(mk)(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
/* t * v  (vec3.h:98) */
static inline S(v3) S(scale)(R t, S(v3) v) { return S(mk)(t * v.e[0], t * v.e[1], t * v.e[2]); }
/* v / t = (1/t) * v  (vec3.h:107) */
static inline S(v3) S(divs)(S(v3) v, R t) { return S(scale)((R)1 / t, v); }
static inline S(v3) S(neg)(S(v3) a) { return S(mk)(-a.e[0], -a.e[1], -a.e[2]); }
/* vec3.h:112 */
static inline R S(dot)(S(v3) u, S(v3) v) { return u.e[0] * v.e[0] + u.e[1] * v.e[1] + u.e[2] * v.e[2]; }
/* vec3.h:51 */
static inline R S(len2)(S(v3) a) { return a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]; }
static inline R S(len)(S(v3) a) { return M_SQRT(S(len2)(a)); }
/* vec3.h:117 */
static inline S(v3) S(cross)(S(v3) u, S(v3) v) {
    return S(mk)(u.e[1] * v.e[2] - u.e[2] * v.e[1],
                 u.e[2] * v.e[0] - u.e[0] * v.e[2],
                 u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
/* vec3.h:126 */
static inline S(v3) S(unit)(S(v3) v) { return S(divs)(v, S(len)(v)); }
/* vec3.h:159 */
static inline S(v3) S(reflect)(S(v3) v, S(v3) n) { return S(sub)(v, S(scale)((R)2 * S(dot)(v, n), n)); }
/* vec3.h:163 */
static inline S(v3) S(refract)(S(v3) uv, S(v3) n, R eta) {
    R cos_theta = M_FMIN(S(dot)(S(neg)(uv), n), (R)1.0);
    S(v3) perp = S(scale)(eta, S(add)(uv, S(scale)(cos_theta, n)));
    S(v3) par = S(scale)(-M_SQRT(M_FABS((R)1.0 - S(len2)(perp))), n);
    return S(add)(perp, par);
}

/* ---- keyed random stream ------------------------------------------------ */

typedef struct { uint64_t seed; uint32_t pixel, sample; rto_stream st; } S(rng);

/* open segment `segment` of (pixel, sample) and skip its first `skip` draws */
static inline void S(rng_seg)(S(rng)* g, uint32_t segment, uint32_t skip) {
    rto_stream_begin(&g->st, g->seed, g->pixel, g->sample, segment);
    for (uint32_t i = 0; i < skip; i++) (void)rto_stream_next(&g->st);
}

/* utils.h:25 random_double(): k24 / 2^24 in [0,1) */
static inline R S(rnd)(S(rng)* g) {
    return (R)rto_stream_next(&g->st) * (R)(1.0 / 16777216.0);
}
/* utils.h:26 */
static inline R S(rnd_range)(S(rng)* g, R lo, R hi) { return lo + (hi - lo) * S(rnd)(g); }

/* vec3.h:130.  vec3::random(-1,1) is vec3(random_double(..), random_double(..),
 * random_double(..)) (vec3.h:60); g++ evaluates constructor arguments right to left, so the
 * first draw lands in z, the second in y, the third in x. */
static inline S(v3) S(in_unit_sphere)(S(rng)* g) {
    for (;;) {
        R z = S(rnd_range)(g, -1, 1);
        R y = S(rnd_range)(g, -1, 1);
        R x = S(rnd_range)(g, -1, 1);
        S(v3) p = S(mk)(x, y, z);
        if (!(S(len2)(p) >= 1)) return p;
    }
}
/* vec3.h:139 */
static inline S(v3) S(unit_vector_rnd)(S(rng)* g) { return S(unit)(S(in_unit_sphere)(g)); }
/* vec3.h:150; first draw lands in y, second in x (same argument-order rule). */
static inline S(v3) S(in_unit_disk)(S(rng)* g) {
    for (;;) {
        R y = S(rnd_range)(g, -1, 1);
        R x = S(rnd_range)(g, -1, 1);
        S(v3) p = S(mk)(x, y, 0);
        if (!(S(len2)(p) >= 1)) return p;
    }
}

/* ---- camera (camera.h:19-45) ---------------------------------------------- */

typedef struct { S(v3) origin, llc, horizontal, vertical, u, v, w; R lens_radius; } S(cam);

static S(cam) S(cam_load)(const double* c) {
    S(cam) k;
    S(v3)* f[7] = { &k.origin, &k.llc, &k.horizontal, &k.vertical, &k.u, &k.v, &k.w };
    for (int i = 0; i < 7; i++) for (int a = 0; a < 3; a++) f[i]->e[a] = (R)c[3 * i + a];
    k.lens_radius = (R)c[21];
    return k;
}

void S(rto_camera_init)(const double* lookfrom, const double* lookat, const double* vup,
                        double vfov, double aspect, double aperture, double focus_dist,
                        double* out) {
    const R pi = (R)3.1415926535897932385;               /* utils.h:14 */
    R theta = (R)vfov * pi / (R)180.0;                   /* utils.h:16 */
    R h = M_TAN(theta / 2);
    R viewport_height = (R)2.0 * h;
    R viewport_width = (R)aspect * viewport_height;
    S(v3) from = S(mk)((R)lookfrom[0], (R)lookfrom[1], (R)lookfrom[2]);
    S(v3) at = S(mk)((R)lookat[0], (R)lookat[1], (R)lookat[2]);
    S(v3) up = S(mk)((R)vup[0], (R)vup[1], (R)vup[2]);
    S(v3) w = S(unit)(S(sub)(from, at));
    S(v3) u = S(unit)(S(cross)(up, w));
    S(v3) v = S(cross)(w, u);
    S(v3) horizontal = S(scale)((R)focus_dist * viewport_width, u);
    S(v3) vertical = S(scale)((R)focus_dist * viewport_height, v);
    S(v3) llc = S(sub)(S(sub)(S(sub)(from, S(divs)(horizontal, 2)), S(divs)(vertical, 2)),
                       S(scale)((R)focus_dist, w));
    R lens_radius = (R)aperture / 2;
    S(v3) f[7] = { from, llc, horizontal, vertical, u, v, w };
    for (int i = 0; i < 7; i++) for (int a = 0; a < 3; a++) out[3 * i + a] = (double)f[i].e[a];
    out[21] = (double)lens_radius;
}

/* camera.h:40 */
static inline void S(get_ray)(const S(cam)* c, R s, R t, S(rng)* g, S(v3)* ro, S(v3)* rd_out) {
    S(v3) rd = S(scale)(c->lens_radius, S(in_unit_disk)(g));
    S(v3) offset = S(add)(S(scale)(rd.e[0], c->u), S(scale)(rd.e[1], c->v));
    *ro = S(add)(c->origin, offset);
    *rd_out = S(sub)(S(sub)(S(add)(S(add)(c->llc, S(scale)(s, c->horizontal)),
                                   S(scale)(t, c->vertical)), c->origin), offset);
}

void S(rto_get_ray)(const double* cam, double s, double t, uint64_t seed, uint32_t pixel,
                    uint32_t sample, uint32_t first_draw, double* ray, uint32_t* draws_used) {
    S(cam) c = S(cam_load)(cam);
    S(rng) g = { seed, pixel, sample, {{0}, {0}, 0, 0, 0} };
    S(rng_seg)(&g, 0, first_draw);
    S(v3) o, d;
    S(get_ray)(&c, (R)s, (R)t, &g, &o, &d);
    for (int a = 0; a < 3; a++) { ray[a] = (double)o.e[a]; ray[3 + a] = (double)d.e[a]; }
    if (draws_used) *draws_used = g.st.d - first_draw;
}

void S(rto_sample)(int which, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment,
                   double* out, uint32_t* draws_used) {
    S(rng) g = { seed, pixel, sample, {{0}, {0}, 0, 0, 0} };
    S(rng_seg)(&g, segment, 0);
    S(v3) v = which == 0 ? S(in_unit_sphere)(&g) : which == 1 ? S(unit_vector_rnd)(&g)
                                                              : S(in_unit_disk)(&g);
    for (int a = 0; a < 3; a++) out[a] = (double)v.e[a];
    if (draws_used) *draws_used = g.st.d;
}

/* ---- closest hit ------------------------------------------------------------- */

typedef struct {
    S(v3) p, normal;
    R t, u, v;
    int front, prim, mat;
} S(hit);

/* sceneobject.h:16 */
static inline void S(set_face_normal)(S(hit)* h, S(v3) dir, S(v3) outward) {
    h->front = S(dot)(dir, outward) < 0;
    h->normal = h->front ? outward : S(neg)(outward);
}

/* sphere.h:18-51; uv from cuda/sphere.h:20-26 on the outward normal */
static int S(sphere_hit)(const double* geom, S(v3) o, S(v3) d, R t_min, R t_max, S(hit)* h) {
    const R pi = (R)3.1415926535897932385;
    S(v3) center = S(mk)((R)geom[0], (R)geom[1], (R)geom[2]);
    R radius = (R)geom[3];
    S(v3) oc = S(sub)(o, center);
    R a = S(len2)(d);
    R b = 2 * S(dot)(oc, d);
    R c = S(dot)(oc, oc) - radius * radius;
    R disc = b * b - 4 * a * c;
    if (disc < 0) return 0;
    R root = (-b - M_SQRT(disc)) / ((R)2.0 * a);
    if (root < t_min || root > t_max) {
        root = (-b + M_SQRT(disc)) / ((R)2.0 * a);
        if (root < t_min || root > t_max) return 0;
    }
    h->t = root;
    h->p = S(add)(o, S(scale)(root, d));
    S(v3) outward = S(divs)(S(sub)(h->p, center), radius);
    S(set_face_normal)(h, d, outward);
    R theta = M_ACOS(-outward.e[1]);
    R phi = M_ATAN2(-outward.e[2], outward.e[0]) + pi;
    h->u = phi / (2 * pi);
    h->v = theta / pi;
    return 1;
}

/* cuda/rect.h:29-52, 75-99, 122-146.  A parallel ray (division by zero) never hits. */
static int S(rect_hit)(int type, const double* geom, S(v3) o, S(v3) d, R t_min, R t_max, S(hit)* h) {
    int ka, aa, ba;  /* plane axis, first and second in-plane axis */
    if (type == RTO_XY_RECT) { ka = 2; aa = 0; ba = 1; }
    else if (type == RTO_XZ_RECT) { ka = 1; aa = 0; ba = 2; }
    else { ka = 0; aa = 1; ba = 2; }
    R a0 = (R)geom[0], a1 = (R)geom[1], b0 = (R)geom[2], b1 = (R)geom[3], k = (R)geom[4];
    R t = (k - o.e[ka]) / d.e[ka];
    if (!(t >= t_min && t <= t_max)) return 0;
    R av = o.e[aa] + t * d.e[aa];
    R bv = o.e[ba] + t * d.e[ba];
    if (!(av >= a0 && av <= a1 && bv >= b0 && bv <= b1)) return 0;
    h->u = (av - a0) / (a1 - a0);
    h->v = (bv - b0) / (b1 - b0);
    h->t = t;
    S(v3) outward = S(mk)(ka == 0, ka == 1, ka == 2);
    S(set_face_normal)(h, d, outward);
    h->p = S(add)(o, S(scale)(t, d));
    return 1;
}

/* cuda/sceneobject.h:80-109 (rotate_y::hit) around a translation: the ray is taken into the
 * instance's frame, the wrapped object is asked, point and normal come back.  Two deliberate
 * differences from the text: sine and cosine are taken in R, not always in float (line 50-52);
 * and the record keeps the side of the surface the wrapped object saw -- line 105 calls
 * set_face_normal with the object-space ray and the world-space normal, which overwrites
 * front_face with "true" at angle 0 and with a mixed-frame sign otherwise. */
static int S(instance_hit)(const rto_scene* sc, int xf, int type, const double* geom, S(v3) o, S(v3) d,
                           R t_min, R t_max, S(hit)* h) {
    const double* x = sc->xform + 4 * (size_t)xf;
    const R pi = (R)3.1415926535897932385;
    R radians = (R)x[0] * pi / (R)180;
    R sn = M_SIN(radians), cs = M_COS(radians);
    S(v3) q = S(sub)(o, S(mk)((R)x[1], (R)x[2], (R)x[3]));
    S(v3) ro = S(mk)(cs * q.e[0] - sn * q.e[2], q.e[1], sn * q.e[0] + cs * q.e[2]);
    S(v3) rd = S(mk)(cs * d.e[0] - sn * d.e[2], d.e[1], sn * d.e[0] + cs * d.e[2]);
    int ok = type == RTO_SPHERE ? S(sphere_hit)(geom, ro, rd, t_min, t_max, h)
                                : S(rect_hit)(type, geom, ro, rd, t_min, t_max, h);
    if (!ok) return 0;
    S(v3) p = h->p, n = h->normal;
    h->p = S(add)(S(mk)(cs * p.e[0] + sn * p.e[2], p.e[1], -sn * p.e[0] + cs * p.e[2]),
                  S(mk)((R)x[1], (R)x[2], (R)x[3]));
    h->normal = S(mk)(cs * n.e[0] + sn * n.e[2], n.e[1], -sn * n.e[0] + cs * n.e[2]);
    return 1;
}

/* sceneobject_list.h:26-39: the later object wins an exact tie because the range test is
 * inclusive. */
static int S(world_hit)(const rto_scene* sc, S(v3) o, S(v3) d, R t_min, R t_max, S(hit)* rec) {
    S(hit) tmp;
    int any = 0;
    R closest = t_max;
    for (int i = 0; i < sc->n_prims; i++) {
        const double* geom = sc->prim_geom + 6 * (size_t)i;
        int type = sc->prim_type[i];
        int xf = sc->n_xforms > 0 && sc->prim_xform ? sc->prim_xform[i] : -1;
        int ok = xf >= 0 ? S(instance_hit)(sc, xf, type, geom, o, d, t_min, closest, &tmp)
               : type == RTO_SPHERE ? S(sphere_hit)(geom, o, d, t_min, closest, &tmp)
                                    : S(rect_hit)(type, geom, o, d, t_min, closest, &tmp);
        if (ok) {
            any = 1;
            closest = tmp.t;
            tmp.prim = i;
            tmp.mat = sc->prim_mat[i];
            *rec = tmp;
        }
    }
    return any;
}

void S(rto_trace)(const rto_scene* sc, int n, const double* rays, double t_min, double t_max,
                  int32_t* id, double* t, double* p, double* nrm, int32_t* front, double* uv) {
    for (int i = 0; i < n; i++) {
        const double* q = rays + 6 * (size_t)i;
        S(v3) o = S(mk)((R)q[0], (R)q[1], (R)q[2]);
        S(v3) d = S(mk)((R)q[3], (R)q[4], (R)q[5]);
        S(hit) h;
        memset(&h, 0, sizeof h);
        if (!S(world_hit)(sc, o, d, (R)t_min, (R)t_max, &h)) {
            id[i] = -1; t[i] = 0; front[i] = 0;
            for (int a = 0; a < 3; a++) { p[3 * i + a] = 0; nrm[3 * i + a] = 0; }
            if (uv) { uv[2 * i] = 0; uv[2 * i + 1] = 0; }
            continue;
        }
        id[i] = h.prim; t[i] = (double)h.t; front[i] = h.front;
        for (int a = 0; a < 3; a++) { p[3 * i + a] = (double)h.p.e[a]; nrm[3 * i + a] = (double)h.normal.e[a]; }
        if (uv) { uv[2 * i] = (double)h.u; uv[2 * i + 1] = (double)h.v; }
    }
}

/* ---- textures (cuda/texture.h) ---------------------------------------------------- */

static S(v3) S(tex_value)(const rto_scene* sc, int tex, R u, R v, S(v3) p) {
    /* checker nesting (cuda/texture.h:43-52) resolved iteratively */
    for (int guard = 0; guard < 16; guard++) {
        int type = sc->tex_type[tex];
        if (type == RTO_TEX_SOLID) {
            const double* c = sc->tex_color + 3 * (size_t)tex;
            return S(mk)((R)c[0], (R)c[1], (R)c[2]);
        }
        if (type == RTO_TEX_CHECKER) {
            R sines = M_SIN((R)10.0 * p.e[0]) * M_SIN((R)10.0 * p.e[1]) * M_SIN((R)10.0 * p.e[2]);
            tex = sines < 0 ? sc->tex_b[tex] : sc->tex_a[tex];
            continue;
        }
        /* image (cuda/texture.h:63-76): v flipped, texel column = uint(u*W); the three byte
         * fetches of the 3W-wide one-channel array land on bytes 3*col+{0,1,2}; both
         * coordinates wrap (cuda/main.cu:421-425: normalised, wrap, point filter). */
        int W = sc->tex_a[tex], H = sc->tex_b[tex];
        const uint8_t* img = sc->texels + sc->tex_off[tex];
        R vf = (R)1.0 - v;
        uint32_t col = (uint32_t)(u * (R)W);
        S(v3) out;
        for (int c = 0; c < 3; c++) {
            /* tex2D coordinate as the reference forms it, in float like the original */
            float xn = (3.0f * (float)col + 0.5f + (float)c) * (1.0f / (float)(3 * W));
            float xw = xn - floorf(xn);
            int ix = (int)floorf(xw * (float)(3 * W));
            if (ix >= 3 * W) ix = 3 * W - 1;
            float yf = (float)vf;
            float yw = yf - floorf(yf);
            int iy = (int)floorf(yw * (float)H);
            if (iy >= H) iy = H - 1;
            out.e[c] = (R)img[(size_t)iy * 3 * W + ix] / (R)255.0;
        }
        return out;
    }
    return S(mk)(0, 0, 0);
}

void S(rto_texture_value)(const rto_scene* sc, int tex, double u, double v, const double* p,
                          double* rgb) {
    S(v3) c = S(tex_value)(sc, tex, (R)u, (R)v, S(mk)((R)p[0], (R)p[1], (R)p[2]));
    for (int a = 0; a < 3; a++) rgb[a] = (double)c.e[a];
}

static S(v3) S(mat_color)(const rto_scene* sc, int mat, const S(hit)* h) {
    int tex = sc->mat_tex[mat];
    if (tex >= 0) return S(tex_value)(sc, tex, h->u, h->v, h->p);
    const double* a = sc->mat_param + 5 * (size_t)mat;
    return S(mk)((R)a[0], (R)a[1], (R)a[2]);
}

/* ---- materials (material.h:15-92; emitters cuda/material.h:33-47,103-119) ----------- */

/* material.h:86 */
static inline R S(reflectance)(R cosine, R r_idx) {
    R r0 = (1 - r_idx) / (1 + r_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * M_POW((1 - cosine), 5);
}

/* returns 1 when the ray scatters; emitted is always written */
static int S(scatter)(const rto_scene* sc, int mat, S(v3) rdir, const S(hit)* h, S(rng)* g,
                      S(v3)* att, S(v3)* so, S(v3)* sd, S(v3)* emitted) {
    int type = sc->mat_type[mat];
    const double* prm = sc->mat_param + 5 * (size_t)mat;
    *emitted = S(mk)(0, 0, 0);
    *so = h->p;
    switch (type) {
    case RTO_LAMBERTIAN: {
        S(v3) dir = S(add)(h->normal, S(unit_vector_rnd)(g));
        const R thr = (R)(float)1e-8;                      /* vec3.h:65 float threshold */
        if (M_FABS(dir.e[0]) < thr && M_FABS(dir.e[1]) < thr && M_FABS(dir.e[2]) < thr)
            dir = h->normal;
        *sd = dir;
        *att = S(mat_color)(sc, mat, h);
        return 1;
    }
    case RTO_METAL: {
        R fuzz = (R)prm[3] < 1 ? (R)prm[3] : 1;            /* material.h:40 */
        S(v3) reflected = S(reflect)(S(unit)(rdir), h->normal);
        *sd = S(add)(reflected, S(scale)(fuzz, S(in_unit_sphere)(g)));
        *att = S(mk)((R)prm[0], (R)prm[1], (R)prm[2]);
        return S(dot)(*sd, h->normal) > 0;
    }
    case RTO_DIELECTRIC: {
        R ir = (R)prm[4];
        *att = S(mk)(1, 1, 1);
        R ratio = h->front ? ((R)1.0 / ir) : ir;
        S(v3) ud = S(unit)(rdir);
        R cos_theta = M_FMIN(S(dot)(S(neg)(ud), h->normal), (R)1.0);
        R sin_theta = M_SQRT((R)1.0 - cos_theta * cos_theta);
        int tir = ratio * sin_theta > (R)1.0;
        if (tir || S(reflectance)(cos_theta, ratio) > S(rnd)(g)) *sd = S(reflect)(ud, h->normal);
        else *sd = S(refract)(ud, h->normal, ratio);
        return 1;
    }
    default:  /* diffuse_light, lambertian_bg: emit their texture, never scatter */
        *emitted = S(mat_color)(sc, mat, h);
        *att = S(mk)(0, 0, 0);
        *sd = S(mk)(0, 0, 0);
        return 0;
    }
}

int S(rto_scatter)(const rto_scene* sc, int mat, const double* ray_in, const double* p,
                   const double* nrm, int front, const double* uv, uint64_t seed,
                   uint32_t pixel, uint32_t sample, uint32_t segment, double* out) {
    S(hit) h;
    memset(&h, 0, sizeof h);
    h.p = S(mk)((R)p[0], (R)p[1], (R)p[2]);
    h.normal = S(mk)((R)nrm[0], (R)nrm[1], (R)nrm[2]);
    h.front = front;
    if (uv) { h.u = (R)uv[0]; h.v = (R)uv[1]; }
    S(rng) g = { seed, pixel, sample, {{0}, {0}, 0, 0, 0} };
    S(rng_seg)(&g, segment, 0);
    S(v3) att, so, sd, em;
    int ok = S(scatter)(sc, mat, S(mk)((R)ray_in[3], (R)ray_in[4], (R)ray_in[5]), &h, &g,
                        &att, &so, &sd, &em);
    for (int a = 0; a < 3; a++) {
        out[a] = (double)att.e[a]; out[3 + a] = (double)so.e[a];
        out[6 + a] = (double)sd.e[a]; out[9 + a] = (double)em.e[a];
    }
    return ok;
}

/* ---- integrator (main.cc:10-35) ---------------------------------------------------- */

/* The recursion  L = attenuation * ray_color(scattered)  is unrolled into a forward pass that
 * records (emitted, attenuation) per bounce and a backward fold in the recursion's own
 * order, with the standard emissive term  L = Le + attenuation * L_next  for the CUDA tree's
 * light materials.  Segment k+1 holds the draws made after the k-th closest-hit query. */
static S(v3) S(ray_color)(const rto_scene* sc, S(v3) o, S(v3) d, int depth, R t_min,
                          S(rng)* g, uint64_t* rays) {
    S(v3) att_stack[depth > 0 ? depth : 1];
    S(v3) emit_stack[depth > 0 ? depth : 1];
    int n = 0;
    S(v3) L = S(mk)(0, 0, 0);   /* depth exhausted: main.cc:13 */
    for (int k = 0; k < depth; k++) {
        S(hit) h;
        memset(&h, 0, sizeof h);
        S(rng_seg)(g, (uint32_t)(k + 1), 0);
        (*rays)++;
        if (S(world_hit)(sc, o, d, t_min, (R)INFINITY, &h)) {
            S(v3) att, so, sd, em;
            if (S(scatter)(sc, h.mat, d, &h, g, &att, &so, &sd, &em)) {
                att_stack[n] = att; emit_stack[n] = em; n++;
                o = so; d = sd;
                continue;
            }
            L = em;   /* absorbed: black (main.cc:26) or the light itself */
            break;
        }
        if (sc->background == RTO_BG_SKY) {                 /* main.cc:30-34 */
            S(v3) ud = S(unit)(d);
            R t = (R)0.5 * (ud.e[1] + (R)1.0);
            L = S(add)(S(scale)((R)1.0 - t, S(mk)(1.0, 1.0, 1.0)),
                       S(scale)(t, S(mk)((R)0.5, (R)0.7, (R)1.0)));
        } else {
            L = S(mk)((R)sc->bg_color[0], (R)sc->bg_color[1], (R)sc->bg_color[2]);
        }
        break;
    }
    for (int k = n - 1; k >= 0; k--) L = S(add)(emit_stack[k], S(mulv)(att_stack[k], L));
    return L;
}

void S(rto_ray_color)(const rto_scene* sc, const double* ray, int depth, double t_min,
                      uint64_t seed, uint32_t pixel, uint32_t sample, double* rgb,
                      uint64_t* rays) {
    S(rng) g = { seed, pixel, sample, {{0}, {0}, 0, 0, 0} };
    uint64_t nr = 0;
    S(v3) c = S(ray_color)(sc, S(mk)((R)ray[0], (R)ray[1], (R)ray[2]),
                           S(mk)((R)ray[3], (R)ray[4], (R)ray[5]), depth, (R)t_min, &g, &nr);
    for (int a = 0; a < 3; a++) rgb[a] = (double)c.e[a];
    if (rays) *rays = nr;
}

/* main.cc:111-126 over the window [i0,i1) x [j0,j1) (every istep-th column, jstep-th row).
 * Pixel index = j*W + i, j counted from the bottom row.  sums: H x W x 3 doubles. */
uint64_t S(rto_render)(const rto_scene* sc, const double* camd, int W, int H, int spp,
                       int sample_begin, int depth, double t_min, uint64_t seed,
                       int i0, int i1, int j0, int j1, int istep, int jstep,
                       int threads, double* sums) {
    S(cam) cam = S(cam_load)(camd);
    uint64_t total = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : total)
    for (int j = j0; j < j1; j++) {
        if ((j - j0) % jstep) continue;
        for (int i = i0; i < i1; i++) {
            if ((i - i0) % istep) continue;
            S(v3) acc = S(mk)(0, 0, 0);
            uint32_t pix = (uint32_t)(j * W + i);
            for (int s = sample_begin; s < sample_begin + spp; s++) {
                S(rng) g = { seed, pix, (uint32_t)s, {{0}, {0}, 0, 0, 0} };
                S(rng_seg)(&g, 0, 0);
                R u = ((R)i + S(rnd)(&g)) / (R)(W - 1);
                R v = ((R)j + S(rnd)(&g)) / (R)(H - 1);
                S(v3) o, d;
                S(get_ray)(&cam, u, v, &g, &o, &d);
                uint64_t nr = 0;
                S(v3) c = S(ray_color)(sc, o, d, depth, (R)t_min, &g, &nr);
                total += nr;
                acc = S(add)(acc, c);                      /* vec3.h:30 operator+= */
            }
            for (int a = 0; a < 3; a++) sums[3 * ((size_t)j * W + i) + a] = (double)acc.e[a];
        }
    }
    return total;
}
