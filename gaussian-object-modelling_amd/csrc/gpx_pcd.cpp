// gpx_pcd.cpp -- host-side input path of the GP hot path: PCD reader and the node's data preparation.
//
//   gpx_pcd_read           : pcl::io::loadPCDFile as used at reference src/gp_node.cpp:557
//                            (ascii / binary / binary_compressed = LZF + struct-of-arrays).
//   gpx_node_training_set  : deMeanAndNormalizeData (src/gp_node.cpp:85-117, PCL float arithmetic),
//                            prepareExtData (:793-850, 3 x 5 points on the radius-2 sphere, label 1),
//                            prepareData (:853-888, surface points, label 0) and computeGP's
//                            concatenation (:898-914).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/gpx.h"

namespace {

// liblzf-compatible decompressor (the format PCL's binary_compressed payload uses).
bool lzf_decompress(const unsigned char *src, size_t n, unsigned char *out, size_t out_len)
{
    size_t ip = 0, op = 0;
    while (ip < n) {
        unsigned ctrl = src[ip++];
        if (ctrl < 32) {
            size_t len = ctrl + 1;
            if (ip + len > n || op + len > out_len)
                return false;
            std::memcpy(out + op, src + ip, len);
            ip += len;
            op += len;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) {
                if (ip >= n)
                    return false;
                len += src[ip++];
            }
            if (ip >= n)
                return false;
            size_t dist = ((size_t)(ctrl & 0x1f) << 8) + src[ip++] + 1;
            len += 2;
            if (dist > op || op + len > out_len)
                return false;
            size_t ref = op - dist;
            for (size_t i = 0; i < len; ++i)
                out[op++] = out[ref++];
        }
    }
    return op == out_len;
}

struct Field {
    std::string name;
    int size = 4;
    char type = 'F';
    int count = 1;
};

double read_scalar(const unsigned char *p, const Field &f)
{
    switch (f.type) {
    case 'F':
        if (f.size == 4) {
            float v;
            std::memcpy(&v, p, 4);
            return v;
        } else {
            double v;
            std::memcpy(&v, p, 8);
            return v;
        }
    case 'U':
        if (f.size == 1)
            return *p;
        if (f.size == 2) {
            uint16_t v;
            std::memcpy(&v, p, 2);
            return v;
        } else {
            uint32_t v;
            std::memcpy(&v, p, 4);
            return v;
        }
    default:
        if (f.size == 1)
            return *(const int8_t *)p;
        if (f.size == 2) {
            int16_t v;
            std::memcpy(&v, p, 2);
            return v;
        } else {
            int32_t v;
            std::memcpy(&v, p, 4);
            return v;
        }
    }
}

}  // namespace

extern "C" long gpx_pcd_read(const char *path, float *xyz, size_t capacity_points)
{
    if (!path)
        return GPX_E_NULL;
    FILE *fh = std::fopen(path, "rb");
    if (!fh)
        return GPX_E_BAD_ARG;
    std::vector<unsigned char> raw;
    {
        unsigned char buf[65536];
        size_t r;
        while ((r = std::fread(buf, 1, sizeof(buf), fh)) > 0)
            raw.insert(raw.end(), buf, buf + r);
        std::fclose(fh);
    }
    std::vector<Field> fields;
    long npts = -1;
    std::string mode;
    size_t pos = 0;
    while (pos < raw.size()) {
        size_t end = pos;
        while (end < raw.size() && raw[end] != '\n')
            ++end;
        std::string line((const char *)raw.data() + pos, end - pos);
        pos = end + 1;
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        if (line.empty() || line[0] == '#')
            continue;
        std::istringstream is(line);
        std::string key;
        is >> key;
        std::vector<std::string> vals;
        for (std::string v; is >> v;)
            vals.push_back(v);
        if (key == "FIELDS") {
            fields.resize(vals.size());
            for (size_t i = 0; i < vals.size(); ++i)
                fields[i].name = vals[i];
        } else if (key == "SIZE") {
            for (size_t i = 0; i < vals.size() && i < fields.size(); ++i)
                fields[i].size = std::atoi(vals[i].c_str());
        } else if (key == "TYPE") {
            for (size_t i = 0; i < vals.size() && i < fields.size(); ++i)
                fields[i].type = vals[i][0];
        } else if (key == "COUNT") {
            for (size_t i = 0; i < vals.size() && i < fields.size(); ++i)
                fields[i].count = std::atoi(vals[i].c_str());
        } else if (key == "POINTS") {
            npts = vals.empty() ? -1 : std::atol(vals[0].c_str());
        } else if (key == "DATA") {
            mode = vals.empty() ? "" : vals[0];
            break;
        }
    }
    if (npts < 0 || fields.empty() || mode.empty())
        return GPX_E_BAD_ARG;
    int ix = -1, iy = -1, iz = -1;
    for (size_t i = 0; i < fields.size(); ++i) {
        if (fields[i].name == "x")
            ix = (int)i;
        if (fields[i].name == "y")
            iy = (int)i;
        if (fields[i].name == "z")
            iz = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0)
        return GPX_E_BAD_ARG;
    if (!xyz)
        return npts;
    if ((size_t)npts > capacity_points)
        return GPX_E_SIZE_MISMATCH;
    const size_t n = (size_t)npts;
    if (mode == "ascii") {
        std::istringstream is(std::string((const char *)raw.data() + pos, raw.size() - pos));
        for (size_t p = 0; p < n; ++p)
            for (size_t f = 0; f < fields.size(); ++f)
                for (int c = 0; c < fields[f].count; ++c) {
                    double v;
                    if (!(is >> v))
                        return GPX_E_BAD_ARG;
                    if (c == 0) {
                        if ((int)f == ix)
                            xyz[3 * p] = (float)v;
                        if ((int)f == iy)
                            xyz[3 * p + 1] = (float)v;
                        if ((int)f == iz)
                            xyz[3 * p + 2] = (float)v;
                    }
                }
        return npts;
    }
    size_t rec = 0;
    std::vector<size_t> foff(fields.size());
    for (size_t f = 0; f < fields.size(); ++f) {
        foff[f] = rec;
        rec += (size_t)fields[f].size * fields[f].count;
    }
    if (mode == "binary") {
        if (pos + rec * n > raw.size())
            return GPX_E_BAD_ARG;
        const unsigned char *d = raw.data() + pos;
        for (size_t p = 0; p < n; ++p) {
            xyz[3 * p] = (float)read_scalar(d + p * rec + foff[ix], fields[ix]);
            xyz[3 * p + 1] = (float)read_scalar(d + p * rec + foff[iy], fields[iy]);
            xyz[3 * p + 2] = (float)read_scalar(d + p * rec + foff[iz], fields[iz]);
        }
        return npts;
    }
    if (mode == "binary_compressed") {
        if (pos + 8 > raw.size())
            return GPX_E_BAD_ARG;
        uint32_t csize, usize;
        std::memcpy(&csize, raw.data() + pos, 4);
        std::memcpy(&usize, raw.data() + pos + 4, 4);
        if (pos + 8 + csize > raw.size() || usize < rec * n)
            return GPX_E_BAD_ARG;
        std::vector<unsigned char> data(usize);
        if (!lzf_decompress(raw.data() + pos + 8, csize, data.data(), usize))
            return GPX_E_BAD_ARG;
        // struct of arrays: all values of field 0, then field 1, ...
        std::vector<size_t> soa(fields.size());
        size_t off = 0;
        for (size_t f = 0; f < fields.size(); ++f) {
            soa[f] = off;
            off += (size_t)fields[f].size * fields[f].count * n;
        }
        for (size_t p = 0; p < n; ++p) {
            xyz[3 * p] = (float)read_scalar(data.data() + soa[ix] + p * fields[ix].size * fields[ix].count, fields[ix]);
            xyz[3 * p + 1] =
                (float)read_scalar(data.data() + soa[iy] + p * fields[iy].size * fields[iy].count, fields[iy]);
            xyz[3 * p + 2] =
                (float)read_scalar(data.data() + soa[iz] + p * fields[iz].size * fields[iz].count, fields[iz]);
        }
        return npts;
    }
    return GPX_E_BAD_ARG;
}

extern "C" int gpx_node_training_set(const float *xyz, size_t n, double sigma2, double rad, double *x, double *y,
                                     double *z, double *label, double *s2)
{
    if (!xyz || !x || !y || !z || !label || !s2)
        return GPX_E_NULL;
    if (n == 0)
        return GPX_E_EMPTY;
    // pcl::compute3DCentroid<PointT,float>: sequential float accumulation, then / n
    float cx = 0, cy = 0, cz = 0;
    for (size_t i = 0; i < n; ++i) {
        cx += xyz[3 * i];
        cy += xyz[3 * i + 1];
        cz += xyz[3 * i + 2];
    }
    cx /= (float)n;
    cy /= (float)n;
    cz /= (float)n;
    std::vector<float> t(3 * n);
    double scale = 0.0;
    for (size_t i = 0; i < n; ++i) {  // demeanPointCloud, then the max-norm scan of :103-108
        float a = xyz[3 * i] - cx, b = xyz[3 * i + 1] - cy, c = xyz[3 * i + 2] - cz;
        t[3 * i] = a;
        t[3 * i + 1] = b;
        t[3 * i + 2] = c;
        float n2 = a * a + b * b;
        n2 = n2 + c * c;
        double norm = std::sqrt(n2);  // std::sqrt(float) -> float, widened
        if (norm >= scale)
            scale = norm;
    }
    const float s = (float)(1.0 / scale);  // Matrix4f sc << 1/current_scale_ ...
    for (size_t i = 0; i < n; ++i) {       // transformPointCloud with a float matrix
        x[i] = (double)(float)(s * t[3 * i]);
        y[i] = (double)(float)(s * t[3 * i + 1]);
        z[i] = (double)(float)(s * t[3 * i + 2]);
        label[i] = 0.0;
        s2[i] = sigma2;
    }
    // prepareExtData: the same accumulating double loops as :821-849
    const int ang_div = 5, lin_div = 3;
    const double ang_step = M_PI * 2 / ang_div;
    const double lin_step = 2 * rad / lin_div;
    size_t k = n;
    for (double lin = -rad + lin_step / 2; lin < rad; lin += lin_step)
        for (double ang = 0; ang < 2 * M_PI; ang += ang_step) {
            if (k >= n + 15)
                return GPX_E_SIZE_MISMATCH;
            x[k] = std::sqrt(std::pow(rad, 2) - lin * lin) * std::cos(ang);
            y[k] = std::sqrt(std::pow(rad, 2) - lin * lin) * std::sin(ang);
            z[k] = lin;
            label[k] = 1.0;
            s2[k] = sigma2;
            ++k;
        }
    return (int)(k - n);  // number of exterior points appended (15)
}
