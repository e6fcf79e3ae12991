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
_DE_CHARS = re.compile(r'[äöüßÄÖÜ]')
_DE_WORDS = re.compile(r'\b(der|die|das|den|dem|des|ein|eine|einen|einem|einer|und|ist|sind|mit|nicht|auch|auf|für|von|zu|im|ich|du|er|sie|es|'
                       r'wir|ihr|mein|dein|sein|guten|tag|danke|bitte|heute|morgen)\b', re.IGNORECASE)


def contains_chinese(text):
    return bool(_ZH.search(text))


def contains_french(text):
    """French letters, or at least two of the common French function words (frontend_utils.py:28-40)."""
    return bool(_FR_CHARS.search(text)) or len(_FR_WORDS.findall(text.lower())) >= 2


def contains_german(text):
    return bool(_DE_CHARS.search(text)) or len(_DE_WORDS.findall(text.lower())) >= 2


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
_DE_SYMBOLS = (('&', ' und '), ('@', ' at '), ('%', ' Prozent '), ('€', ' Euro '), ('$', ' Dollar '), ('°', ' Grad '), ('+', ' plus '),
               ('=', ' gleich '))


def replace_symbols_french(text):
    for a, b in _FR_SYMBOLS:
        text = text.replace(a, b)
    return text


def expand_abbreviations_french(text):
    for pat, rep in _FR_ABBR:
        text = re.sub(pat, rep, text, flags=re.IGNORECASE)
    return text


def spell_out_number_lang(text, lang):
    """Stand-alone integers -> words through num2words when it is installed (frontend_utils.py:75-89); unchanged otherwise."""
    try:
        import num2words
    except ImportError:
        return text
    return re.sub(r'\b\d+\b', lambda m: num2words.num2words(int(m.group()), lang=lang), text)


def replace_symbols_german(text):
    for a, b in _DE_SYMBOLS:
        text = text.replace(a, b)
    return text


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


def normalize_sentence(s, lang):
    """The dependency-free branches of `_normalize_sentence` (cli/frontend.py:335-417): the NeMo / WeTextProcessing / ttsfrd
    normalisers are optional in the reference too and are not installed here."""
    s = s.replace('\n', ' ').strip()
    if lang == 'fr':
        s = replace_symbols_french(spell_out_number_lang(expand_abbreviations_french(s), 'fr'))
    elif lang == 'de':
        s = replace_symbols_german(spell_out_number_lang(s, 'de'))
    elif lang == 'en':
        s = spell_out_number_lang(s, 'en')
    else:
        return s
    return re.sub(r'\s+', ' ', remove_bracket(s)).strip()
