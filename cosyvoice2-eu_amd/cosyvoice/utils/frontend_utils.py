"""Text-side helpers of the frontend (cosyvoice/utils/frontend_utils.py of the reference): paragraph splitting by token
budget, punctuation-only filter, light language heuristics and the dependency-free French / German fallbacks of
`_normalize_sentence` (cli/frontend.py:335-417).  Host-only string work; pinned by tests/golden/split_paragraph.json, which
holds the reference functions' outputs on sample paragraphs.
"""
import re

import regex

_ZH = re.compile(r'[一-鿿]+')
_FR_CHARS = re.compile(r'[àâäéèêëïîôùûüÿç]')
_FR_WORDS = re.compile(r'\b(le|la|les|un|une|des|du|de|et|est|avec|dans|pour|sur|par|ce|cette|qui|que|dont|où|si|mais|ou|donc|car|ni|or|je|tu|il|'
                       r'elle|nous|vous|ils|elles|mon|ma|mes|ton|ta|tes|son|sa|ses|notre|votre|leur|leurs|bonjour|bonsoir|merci|salut|'
                       r'français|habite|appelle|travaille)\b', re.IGNORECASE)
_DE_CHARS = re.compile(r'[äöüÄÖÜß]')
# the reference imports its German helpers from frontend_utils, where they do not exist, so its `_fallback_*` functions are what
# runs (cli/frontend.py:46-57, 64-73): ONE of these words is enough
_DE_WORDS = re.compile(r'\b(und|oder|nicht|mit|ist|ein|eine|der|die|das|zum|beispiel|bzw|genau|genommen|seit|schon|bereits|heute|gestern|morgen|'
                       r'wird|wurden?|kann|können|deutsch|spr[eä]che?)\b', re.IGNORECASE)


def contains_chinese(text):
    return bool(_ZH.search(text))


def contains_french(text):
    """French letters, or at least two of the common French function words (frontend_utils.py:28-40)."""
    return bool(_FR_CHARS.search(text)) or len(_FR_WORDS.findall(text.lower())) >= 2


def contains_german(text):
    """cli/frontend.py:64-73 (`_fallback_contains_german`): a German letter, or any one of its signal words."""
    return bool(_DE_CHARS.search(text)) or bool(_DE_WORDS.search(text))


def remove_bracket(text):
    for a in ('（', '）', '【', '】', '`'):
        text = text.replace(a, '')
    return text.replace('——', ' ')


def replace_corner_mark(text):
    return text.replace('²', '平方').replace('³', '立方')


def replace_blank(text):
    """Drop blanks unless both neighbours are non-blank ASCII (blanks between Chinese characters go)."""
    out = []
    for i, c in enumerate(text):
        if c != ' ':
            out.append(c)
        elif 0 < i < len(text) - 1 and text[i + 1].isascii() and text[i + 1] != ' ' and text[i - 1].isascii() and text[i - 1] != ' ':
            out.append(c)
    return ''.join(out)


_FR_SYMBOLS = (('&', ' et '), ('@', ' arobase '), ('%', ' pour cent '), ('#', ' dièse '), ('$', ' dollar '), ('€', ' euros '),
               ('£', ' livres '), ('°', ' degrés '), ('+', ' plus '), ('=', ' égal '))
_FR_ABBR = ((r'\bM\.', 'monsieur'), (r'\bMme\.?', 'madame'), (r'\bMlle\.?', 'mademoiselle'), (r'\bDr\.', 'docteur'), (r'\bPr\.', 'professeur'),
            (r'\bSt\.', 'saint'), (r'\bCie\.?', 'compagnie'), (r'\betc\.', 'et cetera'), (r'\bc-à-d\.?', "c'est-à-dire"),
            (r'\bp\.ex\.', 'par exemple'), (r'\bav\.', 'avenue'), (r'\bbd\.?', 'boulevard'), (r'\bpl\.', 'place'), (r'\brue\.', 'rue'))
_DE_ABBR = ((r'\bz\.?\s?B\.?\b', 'zum Beispiel'), (r'\bu\.?\s?a\.?\b', 'unter anderem'), (r'\bbzw\.?\b', 'beziehungsweise'), (r'\bca\.?\b', 'circa'),
            (r'\bd\.?\s?h\.?\b', 'das heißt'), (r'\binsb\.?\b', 'insbesondere'), (r'\bNr\.?\b', 'Nummer'), (r'\bS\.?\b', 'Seite'))
_DE_DIGITS = {'0': 'null', '1': 'eins', '2': 'zwei', '3': 'drei', '4': 'vier', '5': 'fünf', '6': 'sechs', '7': 'sieben', '8': 'acht', '9': 'neun'}


def replace_symbols_french(text):
    for a, b in _FR_SYMBOLS:
        text = text.replace(a, b)
    return text


def expand_abbreviations_french(text):
    for pat, rep in _FR_ABBR:
        text = re.sub(pat, rep, text, flags=re.IGNORECASE)
    return text


def _num2words():
    try:
        from num2words import num2words
        return num2words
    except Exception:
        return None


def spell_out_number_french(text):
    """frontend_utils.py:75-89: stand-alone integers -> French words through num2words; unchanged when it is not installed."""
    n2w = _num2words()
    if n2w is None:
        return text
    return re.sub(r'\b\d+\b', lambda m: n2w(int(m.group()), lang='fr'), text)


def expand_abbreviations_german(text):
    """cli/frontend.py:75-89 (`_fallback_expand_abbreviations_german`)."""
    for pat, rep in _DE_ABBR:
        text = re.sub(pat, rep, text, flags=re.IGNORECASE)
    return text


def replace_symbols_german(text):
    """cli/frontend.py:91-100 (`_fallback_replace_symbols_german`): note that it collapses whitespace and strips itself."""
    text = text.replace('€', ' Euro ').replace('%', ' Prozent ')
    text = re.sub(r'\bkm/?h\b', ' Kilometer pro Stunde ', text, flags=re.IGNORECASE)
    text = text.replace('&', ' und ').replace('@', ' at ').replace('§', ' Paragraph ').replace('°C', ' Grad Celsius ')
    return re.sub(r'\s+', ' ', text).strip()


def spell_out_number_german(text):
    """cli/frontend.py:102-140 (`_fallback_spell_out_number_german`): "<n>." before whitespace / end -> ordinal, decimals with a
    comma -> "<int> Komma <digit words>", then grouped and plain integers.  Without num2words (optional in the reference too) the
    integer parts stay digits; what still changes is the shape: thousands separators go, the fraction is spelled digit by digit."""
    n2w = _num2words()

    def words(n, **kw):
        if n2w is not None:
            try:
                return n2w(n, lang='de', **kw)
            except Exception:
                return None
        return None

    def ord_repl(m):
        n = int(m.group(1))
        return words(n, to='ordinal') or f'{n}.'
    text = re.sub(r'\b(\d+)\.(?=\s|$)', ord_repl, text)

    def dec_repl(m):
        intp, frac = m.group(0).replace('.', '').replace(' ', '').split(',', 1)
        left = words(int(intp)) or intp
        return f"{left} Komma {' '.join(_DE_DIGITS.get(ch, ch) for ch in frac)}"
    text = re.sub(r'\b\d{1,3}(?:[.\s]\d{3})*,\d+\b', dec_repl, text)

    def int_repl(m):
        d = m.group(0).replace('.', '').replace(' ', '')
        return words(int(d)) or d
    text = re.sub(r'\b\d{1,3}(?:[.\s]\d{3})+\b', int_repl, text)
    return re.sub(r'\b\d+\b', int_repl, text)


class NumberWords:
    """`inflect.engine().number_to_words(str)` for a run of digits with its default arguments (group 0, "and", comma ","), the one
    call the reference makes (frontend_utils.py:57-73 via cli/frontend.py:411-415).  inflect (a hard import of the reference) is
    not installed here: this follows its published algorithm -- three-digit groups from the right through the same regular
    expressions and helper strings -- and is PARITY-UNPINNED against the package itself."""
    UNIT = ['', 'one', 'two', 'three', 'four', 'five', 'six', 'seven', 'eight', 'nine']
    TEEN = ['ten', 'eleven', 'twelve', 'thirteen', 'fourteen', 'fifteen', 'sixteen', 'seventeen', 'eighteen', 'nineteen']
    TEN = ['', '', 'twenty', 'thirty', 'forty', 'fifty', 'sixty', 'seventy', 'eighty', 'ninety']
    MILL = ['', ' thousand', ' million', ' billion', ' trillion', ' quadrillion', ' quintillion', ' sextillion', ' septillion', ' octillion',
            ' nonillion', ' decillion']
    _THREE = re.compile(r'(\d)(\d)(\d)(?=\D*\Z)')
    _TWO = re.compile(r'(\d)(\d)(?=\D*\Z)')
    _ONE = re.compile(r'(\d)(?=\D*\Z)')

    def _mill(self, i):
        if i > len(self.MILL) - 1:
            raise ValueError('number out of range')
        return self.MILL[i]

    def _ten(self, tens, units, mindex=0):
        if tens != 1:
            return f"{self.TEN[tens]}{'-' if tens and units else ''}{self.UNIT[units]}{self._mill(mindex)}"
        return f'{self.TEEN[units]}{self.MILL[mindex]}'

    def _hund(self, h, t, u, mindex):
        if h:
            return f"{self.UNIT[h]} hundred{' and ' if t or u else ''}{self._ten(t, u)}{self._mill(mindex)}, "
        if t or u:
            return f'{self._ten(t, u)}{self._mill(mindex)}, '
        return ''

    def _enword(self, num):
        if int(num) == 0:
            return 'zero'
        if int(num) == 1:
            return 'one'
        num = num.lstrip().lstrip('0')
        self._count = 0

        def hundsub(m):
            r = self._hund(int(m.group(1)), int(m.group(2)), int(m.group(3)), self._count)
            self._count += 1
            return r
        while self._THREE.search(num):
            num = self._THREE.sub(hundsub, num, 1)
        num = self._TWO.sub(lambda m: f'{self._ten(int(m.group(1)), int(m.group(2)), self._count)}, ', num, 1)
        return self._ONE.sub(lambda m: f'{self.UNIT[int(m.group(1))]}{self._mill(self._count)}, ', num, 1)

    def number_to_words(self, num):
        chunk = re.sub(r'\D', '', str(num)) or '0'
        chunk = self._enword(chunk)
        if chunk[-2:] == ', ':
            chunk = chunk[:-2]
        chunk = re.sub(r'\s+,', ',', chunk)
        chunk = re.sub(r', (\S+)\s+\Z', r' and \1', chunk)
        return re.sub(r'\s+', ' ', chunk).strip()


def spell_out_number(text, inflect_parser):
    """frontend_utils.py:57-73: EVERY maximal run of digits (not only stand-alone integers) -> inflect_parser.number_to_words(run)."""
    out, st = [], None
    for i, c in enumerate(text):
        if not c.isdigit():
            if st is not None:
                out.append(inflect_parser.number_to_words(text[st:i]))
                st = None
            out.append(c)
        elif st is None:
            st = i
    if st is not None and st < len(text):
        out.append(inflect_parser.number_to_words(text[st:]))
    return ''.join(out)


def split_paragraph(text, tokenize, lang='zh', token_max_n=80, token_min_n=60, merge_len=20, comma_split=False):
    """frontend_utils.py:137-189.  Cut at sentence punctuation (a closing quote stays with its sentence), then pack the pieces
    greedily: a piece starts a new segment when adding it would pass token_max_n and the segment already holds more than
    token_min_n; a final segment shorter than merge_len joins its predecessor.  Length = characters for zh, tokens otherwise."""
    zh = lang == 'zh'
    length = (lambda s: len(s)) if zh else (lambda s: len(tokenize(s)))
    marks = set(['。', '？', '！', '；', '：', '、', '.', '?', '!', ';'] if zh else ['.', '?', '!', ';', ':'])
    if comma_split:
        marks |= {'，', ','}
    if text[-1] not in marks:
        text += '。' if zh else '.'
    pieces, start, i = [], 0, 0
    n = len(text)
    while i < n:
        if text[i] in marks:
            had = i > start
            if had:
                pieces.append(text[start:i + 1])
            if i + 1 < n and text[i + 1] in ('"', '”'):
                # the reference pops the last piece unconditionally here (even when this mark added none)
                last = pieces.pop(-1)
                pieces.append(last + text[i + 1])
                start = i + 2
            else:
                start = i + 1
        i += 1
    out, cur = [], ''
    for p in pieces:
        if length(cur + p) > token_max_n and length(cur) > token_min_n:
            out.append(cur)
            cur = ''
        cur += p
    if cur:
        if length(cur) < merge_len and out:
            out[-1] += cur
        else:
            out.append(cur)
    return out


def is_only_punctuation(text):
    return bool(regex.fullmatch(r'^[\p{P}\p{S}]*$', text))


def split_sentences(text):
    """cli/frontend.py:293-294: cut after . ? ! … 。 ！ ？ when whitespace follows."""
    return [s.strip() for s in re.split(r'(?<=[\.\?\!…。！？])\s+', text) if s.strip()]


def detect_lang(s):
    """cli/frontend.py:296-319 without the optional Lingua detector: Chinese short-circuit, then the FR / DE heuristics."""
    if contains_chinese(s):
        return 'zh'
    if contains_french(s):
        return 'fr'
    if contains_german(s):
        return 'de'
    return 'en'


def inflect_engine():
    """The reference's `inflect.engine()` when the package is installed, else the restatement above."""
    try:
        import inflect
        return inflect.engine()
    except Exception:
        return NumberWords()


def normalize_sentence(s, lang, inflect_parser=None):
    """The dependency-free branches of `_normalize_sentence` (cli/frontend.py:335-417): the NeMo / WeTextProcessing / ttsfrd
    normalisers are optional in the reference too and are not installed here.  French and German: abbreviations -> numbers ->
    symbols -> brackets -> whitespace; English: digit runs through inflect only (no bracket / whitespace pass, frontend.py:411-417);
    Chinese without a normaliser: unchanged."""
    s = s.replace('\n', ' ').strip()
    if lang == 'fr':
        s = replace_symbols_french(spell_out_number_french(expand_abbreviations_french(s)))
        return re.sub(r'\s+', ' ', remove_bracket(s)).strip()
    if lang == 'de':
        s = replace_symbols_german(spell_out_number_german(expand_abbreviations_german(s)))
        return re.sub(r'\s+', ' ', remove_bracket(s)).strip()
    if lang == 'en':
        try:
            return spell_out_number(s, inflect_parser or inflect_engine())
        except Exception:
            return s
    return s
