"""TEST-ONLY: small edits of JPEG marker segments (no re-encoding)."""


def retarget_huffman_tables(data, dc_map, ac_map):
    """The same stream as an EXTENDED sequential frame (SOF1 instead of SOF0: baseline frames may only use ids 0 and 1, the reference
    refuses others in parse_dht) with its Huffman tables stored under other ids: DHT destinations and SOS selectors renamed
    (dc_map / ac_map: old id -> new id, ids 0..3)."""
    out = bytearray(data[:2])
    i = 2
    while i < len(data):
        assert data[i] == 0xFF, i
        m = data[i + 1]
        if m == 0xD9 or 0xD0 <= m <= 0xD7 or m == 0x01:
            out += data[i:i + 2]
            i += 2
            continue
        length = (data[i + 2] << 8) | data[i + 3]
        seg = bytearray(data[i:i + 2 + length])
        if m == 0xC0:
            seg[1] = 0xC1
        elif m == 0xC4:
            p = 4
            while p < len(seg):
                tc, th = seg[p] >> 4, seg[p] & 15
                seg[p] = (tc << 4) | (ac_map if tc else dc_map).get(th, th)
                p += 17 + sum(seg[p + 1:p + 17])
        elif m == 0xDA:
            n = seg[4]
            for k in range(n):
                b = seg[5 + 2 * k + 1]
                seg[5 + 2 * k + 1] = (dc_map.get(b >> 4, b >> 4) << 4) | ac_map.get(b & 15, b & 15)
            out += seg
            out += data[i + 2 + length:]  # entropy-coded data and everything behind it, untouched
            return bytes(out)
        out += seg
        i += 2 + length
    return bytes(out)
