"""Counter-based ("keyed") random numbers shared by every implementation of the Photon path.

The reference draws ``random.uniform(-1, 1)`` three times per bounce and per transmit
(T3DR.py:740, T3DR.py:937) from Python's global Mersenne Twister, in depth-first row-major
order.  That stream is inherently serial, so a parallel renderer cannot reproduce it.  The
keyed generator replaces it by a pure function of *where* the draw happens:

    (seed, pixelX, pixelY, sample, heap index of the ray being generated, component 0..2)

The heap index numbers the nodes of the trace tree (T3DR.py:959-1001): the primary ray is 1,
the bounce child of node n is 2n (``node.left``), the transmit child is 2n+1 (``node.right``).

Every draw carries exactly 24 random bits so that the value is the same number in float32
(HIP kernels) and float64 (reference, oracle).
"""

MASK32 = 0xFFFFFFFF


def mix32(x: int) -> int:
    """32-bit avalanche finaliser ("lowbias32")."""
    x &= MASK32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & MASK32
    x ^= x >> 15
    x = (x * 0x846CA68B) & MASK32
    x ^= x >> 16
    return x


def keyed_bits24(seed: int, px: int, py: int, sample: int, heap: int, comp: int) -> int:
    """24 random bits for one draw."""
    h = mix32((seed + 0x9E3779B9) & MASK32)
    h = mix32(h ^ (px & MASK32))
    h = mix32(h ^ (py & MASK32))
    h = mix32(h ^ (((sample & 0xFFFF) << 16) | (heap & 0xFFFF)))
    h = mix32(h ^ (((comp + 1) * 0x9E3779B9) & MASK32))
    return h >> 8


def keyed_uniform(seed, px, py, sample, heap, comp, a=-1.0, b=1.0) -> float:
    """Same contract as ``random.uniform(a, b)``: ``a + (b - a) * u`` with u in [0, 1)."""
    u = keyed_bits24(seed, px, py, sample, heap, comp) / 16777216.0
    return a + (b - a) * u
