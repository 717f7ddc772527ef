// Dumps what ICU (through node's TextDecoder) decodes, as plain data files next to this script.
// ORACLE-side table source (test infrastructure): the product's tables come from CPython's codecs
// (stringsext_amd/csrc/gen_tables.py); tests/test_tables.py compares the two.
//   node oracle/tables/dump_icu.js        (node 12 / ICU 70.1 in this image)
// Output lines:  <hex bytes> <code point>[+<code point>]   — only sequences that decode without error
// and consume all bytes as ONE unit (a lead followed by an ASCII byte that is passed through is skipped).
const fs = require('fs');
const path = require('path');
function dec(label) { return new TextDecoder(label, { fatal: true }); }
function cps(d, bytes) {
  try { return Array.from(d.decode(Uint8Array.from(bytes))).map(c => c.codePointAt(0)); } catch (e) { return null; }
}
const hex = (n, w) => n.toString(16).padStart(w || 2, '0');
function write(name, lines) { fs.writeFileSync(path.join(__dirname, name), lines.join('\n') + '\n'); }

// single-byte: bytes 0x80..0xFF of every WHATWG single-byte label ICU knows
const SB = ["koi8-r", "ibm866", "iso-8859-2", "iso-8859-5", "iso-8859-15", "windows-1251", "windows-1252", "iso-8859-3",
  "iso-8859-4", "iso-8859-6", "iso-8859-7", "iso-8859-8", "iso-8859-8-i", "iso-8859-10", "iso-8859-13", "iso-8859-14",
  "iso-8859-16", "koi8-u", "macintosh", "windows-874", "windows-1250", "windows-1253", "windows-1254", "windows-1255",
  "windows-1256", "windows-1257", "windows-1258", "x-mac-cyrillic"];
const sb = [];
for (const l of SB) {
  let d;
  try { d = dec(l); } catch (e) { sb.push(l + ' UNSUPPORTED'); continue; }
  const row = [];
  for (let b = 0x80; b <= 0xFF; b++) { const r = cps(d, [b]); row.push(r && r.length == 1 ? hex(r[0], 4) : '0'); }
  sb.push(l + ' ' + row.join(' '));
}
write('icu_single_byte.txt', sb);

function two(label, file, leads, trails) {
  const d = dec(label), out = [];
  for (const l of leads) for (const t of trails) {
    const r = cps(d, [l, t]);
    if (!r) continue;
    if (r.length == 2 && r[1] == t && t < 0x80) continue;  // lead error + ASCII passed through
    out.push(hex(l) + hex(t) + ' ' + r.map(x => hex(x, 4)).join('+'));
  }
  return out;
}
const range = (a, b) => { const v = []; for (let i = a; i <= b; i++) v.push(i); return v; };
write('icu_big5.txt', two('big5', null, range(0x81, 0xFE), range(0x40, 0x7E).concat(range(0xA1, 0xFE))));
const ej = two('euc-jp', null, [0x8E].concat(range(0xA1, 0xFE)), range(0xA1, 0xFE));
{
  const d = dec('euc-jp');
  for (const a of range(0xA1, 0xFE)) for (const b of range(0xA1, 0xFE)) {
    const r = cps(d, [0x8F, a, b]);
    if (r) ej.push('8f' + hex(a) + hex(b) + ' ' + r.map(x => hex(x, 4)).join('+'));
  }
}
write('icu_euc_jp.txt', ej);
write('icu_shift_jis.txt', two('shift_jis', null, range(0x81, 0x9F).concat(range(0xE0, 0xFC)), range(0x40, 0x7E).concat(range(0x80, 0xFC))));
write('icu_euc_kr.txt', two('euc-kr', null, range(0x81, 0xFE), range(0x41, 0xFE)).filter(l => l.indexOf('+') < 0));
console.log('ICU', process.versions.icu, 'Unicode', process.versions.unicode);
// gb18030 (GBK is the same decoder): the two-byte index (leads 81..FE, trails 40..7E and 80..FE) and the four-byte
// sequences below pointer 39420 (the BMP ranges), written as breakpoints: "<pointer> <code point>" where the linear
// run cp = cp0 + (pointer - pointer0) begins; "<pointer> -" where a stretch without mapping begins.
{
  write('icu_gb18030.txt', two('gb18030', null, range(0x81, 0xFE), range(0x40, 0x7E).concat(range(0x80, 0xFE))).filter(l => l.indexOf('+') < 0));
  const d = dec('gb18030'), out = [];
  let prev = null, prevp = -2;
  for (let p = 0; p < 39420; p++) {
    const b = [0x81 + Math.floor(p / 12600), 0x30 + Math.floor(p / 1260) % 10, 0x81 + Math.floor(p / 10) % 126, 0x30 + p % 10];
    const r = cps(d, b);
    const cp = (r && r.length == 1) ? r[0] : null;
    if (cp === null) { if (prev !== null || p == 0) out.push(p + ' -'); }
    else if (prev === null || cp != prev + 1 || prevp != p - 1) out.push(p + ' ' + hex(cp, 4));
    prev = cp; prevp = p;
  }
  write('icu_gb18030_ranges.txt', out);
}
