/*
 * randperm_torch_oracle.c -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone; nothing under probaforms_amd/ touches it).
 *
 * CPU restatement of `torch.randperm(n, generator=g)` on a CPU generator: the epoch shuffle of the reference's
 * DataLoader(shuffle=True) (/root/reference/probaforms/models/realnvp.py:235, cvae.py:235 -> RandomSampler -> randperm), as the
 * build's rnvp_randperm_torch_cpu (probaforms_amd/csrc/rnvp_randperm.hip) restates it for the device:
 *   - the mt19937 engine of ATen/core/MT19937RNGEngine.h (624-word state, tempering), one 32-bit draw per swap;
 *   - ATen/native/TensorFactories.cpp randperm_cpu for n < 2^32 / 20: the identity, then for i = 0 .. n-2
 *         z = generator->random() % (n - i);  swap(r[i], r[i + z]).
 * mt625: the generator's 624 state words + the position of the next unread word (624 = block used up), advanced in place.
 * The reference's arithmetic lives in PyTorch: this restatement is pinned against torch.randperm itself by
 * tests/test_randperm.py; the device path is pinned against torch.randperm directly (tests/test_randperm_gpu.py).
 */
#include <stdint.h>

static void twist(uint32_t *p) {
    int j;
#define MIX(u, v) (((((u) & 0x80000000u) | ((v) & 0x7fffffffu)) >> 1) ^ (((v) & 1u) ? 0x9908b0dfu : 0u))
    for (j = 0; j < 624 - 397; ++j) p[j] = p[j + 397] ^ MIX(p[j], p[j + 1]);
    for (; j < 623; ++j) p[j] = p[j + 397 - 624] ^ MIX(p[j], p[j + 1]);
    p[623] = p[396] ^ MIX(p[623], p[0]);
#undef MIX
}

static uint32_t next_u32(uint32_t *mt625) {
    if (mt625[624] >= 624) { twist(mt625); mt625[624] = 0; }
    uint32_t y = mt625[mt625[624]++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

int randperm_torch(uint32_t *mt625, int64_t n, int64_t *r) {
    if (n < 0 || n >= (int64_t)(0xffffffffu / 20)) return -1;            /* torch shuffles larger n another way */
    for (int64_t i = 0; i < n; ++i) r[i] = i;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const int64_t z = (int64_t)next_u32(mt625) % (n - i);
        const int64_t sav = r[i];
        r[i] = r[z + i];
        r[z + i] = sav;
    }
    return 0;
}
