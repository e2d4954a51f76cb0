"""Counter-based RNG of the device kernels, restated in Python integers (TEST INFRASTRUCTURE).
Must match learninghumanoidwalking_amd/csrc/lhw_rng.h bit for bit."""
STREAM_RESET, STREAM_STEP, STREAM_POLICY = 1, 2, 3
_M = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M
    return x ^ (x >> 31)


def bits(seed: int, env: int, stream: int, counter: int, slot: int) -> int:
    k = splitmix64((seed ^ (0xD1342543DE82EF95 * (env + 1))) & _M)
    return splitmix64(k ^ (stream << 56) ^ (counter << 16) ^ slot)


def u01(seed, env, stream, counter, slot) -> float:
    return float(bits(seed, env, stream, counter, slot) >> 11) * (1.0 / 9007199254740992.0)


def uniform(seed, env, stream, counter, slot, lo, hi) -> float:
    return lo + (hi - lo) * u01(seed, env, stream, counter, slot)


def randint(seed, env, stream, counter, slot, n) -> int:
    r = int(u01(seed, env, stream, counter, slot) * float(n))
    return n - 1 if r >= n else r
