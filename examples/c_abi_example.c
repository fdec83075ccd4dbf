/* c_abi_example.c -- the C ABI of libgpx.so from plain C99 (no C++, no Python): create a model on a small synthetic
 * sphere, evaluate mean / variance / gradient at a few points, project a point onto the surface, append points.
 *
 *   gcc -std=c99 -O2 examples/c_abi_example.c -I include -L gaussian-object-modelling_amd/lib -lgpx \
 *       -Wl,-rpath,$PWD/gaussian-object-modelling_amd/lib -Wl,-rpath-link,/opt/rocm/lib -lm -o c_abi_example
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpx.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != GPX_OK) {                                               \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, gpx_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(void)
{
    enum { N = 200, NEXT = 16, NQ = 5 };
    double x[N + NEXT], y[N + NEXT], z[N + NEXT], label[N + NEXT], sigma2[N + NEXT];
    int i;
    /* surface points of the unit sphere (label 0) and a ring of exterior points at radius 2 (label 1) */
    for (i = 0; i < N + NEXT; ++i) {
        const int outer = (i >= N - 10 && i < N) || i >= N + NEXT - 2;
        const double r = outer ? 2.0 : 1.0;
        const double phi = acos(1.0 - 2.0 * (i + 0.5) / (N + NEXT)), th = 3.14159265358979323846 * (1.0 + sqrt(5.0)) * (i + 0.5);
        x[i] = r * sin(phi) * cos(th), y[i] = r * sin(phi) * sin(th), z[i] = r * cos(phi);
        label[i] = outer ? 1.0 : 0.0;
        sigma2[i] = 0.1;
    }
    if (gpx_device_count() < 1) {
        fprintf(stderr, "no HIP device: libgpx has no CPU path (%s)\n", gpx_version());
        return 2;
    }
    gpx_kernel kern = {GPX_KERNEL_MATERN52, 0, {1.0, 1.0, 0.0, 0.0}};
    gpx_options opt = {0};
    opt.precision = GPX_PREC_F64, opt.device = -1, opt.ir_steps = -1;
    gpx_model *gp = NULL;
    CHECK(gpx_model_create(&kern, N, x, y, z, label, sigma2, &opt, &gp));

    const double qx[NQ] = {0.0, 0.7, 1.0, 1.5, x[3]}, qy[NQ] = {0.0, 0.1, 0.0, 0.2, y[3]}, qz[NQ] = {0.0, -0.3, 0.0, 0.9, z[3]};
    double f[NQ], v[NQ], grad[3 * NQ];
    CHECK(gpx_model_evaluate(gp, NQ, qx, qy, qz, f, v, grad, NULL, NULL));
    for (i = 0; i < NQ; ++i)
        printf("q = (%5.2f %5.2f %5.2f)  f = %+.6f  v = %.6f  |grad| = %.4f\n", qx[i], qy[i], qz[i], f[i], v[i],
               sqrt(grad[3 * i] * grad[3 * i] + grad[3 * i + 1] * grad[3 * i + 1] + grad[3 * i + 2] * grad[3 * i + 2]));

    /* AtlasBase::project, batched: bring a point 20 % outside the surface onto f = 0 along the gradient */
    const double sx = 1.2 * x[5], sy = 1.2 * y[5], sz = 1.2 * z[5];
    double sf, sgrad[3];
    CHECK(gpx_model_evaluate(gp, 1, &sx, &sy, &sz, &sf, NULL, sgrad, NULL, NULL));
    gpx_project_options po = {1e-2, 1e-7, 1.0, 200, {0, 0, 0}};
    double out_xyz[3], out_f;
    int32_t iters, status;
    CHECK(gpx_model_project(gp, 1, &sx, &sy, &sz, sgrad, &po, out_xyz, &out_f, &iters, &status));
    printf("start f = %+.4f; projected to (%.4f %.4f %.4f): f = %+.2e after %d iterations (status %d), radius %.4f\n",
           sf, out_xyz[0], out_xyz[1], out_xyz[2], out_f, (int)iters, (int)status,
           sqrt(out_xyz[0] * out_xyz[0] + out_xyz[1] * out_xyz[1] + out_xyz[2] * out_xyz[2]));

    /* append the remaining points (extends the factor) and look again */
    CHECK(gpx_model_update(gp, NEXT, x + N, y + N, z + N, label + N, sigma2 + N));
    CHECK(gpx_model_evaluate(gp, NQ, qx, qy, qz, f, v, NULL, NULL, NULL));
    printf("after update (+%d points): f(0.7,0.1,-0.3) = %+.6f  v = %.6f\n", NEXT, f[1], v[1]);
    gpx_model_destroy(gp);
    return 0;
}
