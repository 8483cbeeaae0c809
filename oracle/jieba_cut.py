"""Checker for the native dictionary cutter (easyrag_amd/csrc/text.hip): a plain-Python restatement of what
``jieba.Tokenizer.cut(sentence, cut_all=False, HMM=False)`` does in jieba 0.42.1 (the version the reference pins,
/root/reference/requirements.txt:101; used at src/easyrag/pipeline/pipeline.py:176-178 and
src/easyrag/custom/retrievers.py:72-76), written from the published source of that release:
``gen_pfdict`` (prefix dictionary), ``get_DAG``, ``calc`` (maximum log-probability route, ties to the longer word),
``__cut_DAG_NO_HMM`` (single ASCII letters / digits are glued) and the block splitting of ``cut`` (re_han_default,
re_skip_default).  jieba itself is not installable here, so this restatement is **unpinned**: it anchors the C++ against
an independent implementation of the same published algorithm, not against jieba's output.  Test infrastructure only.
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterator, List

re_han_default = re.compile("([一-鿕a-zA-Z0-9+#&\\._%\\-]+)", re.U)
re_skip_default = re.compile("(\r\n|\\s)", re.U)
re_eng = re.compile("[a-zA-Z0-9]", re.U)


class DictCutter:
    def __init__(self, dict_text: str):
        self.FREQ: Dict[str, int] = {}
        self.total = 0
        for line in dict_text.split("\n"):
            line = line.strip()
            if not line:
                continue
            word, freq = line.split(" ")[:2]
            freq = int(freq)
            self.FREQ[word] = freq
            self.total += freq
            for ch in range(len(word)):
                wfrag = word[: ch + 1]
                if wfrag not in self.FREQ:
                    self.FREQ[wfrag] = 0

    def get_DAG(self, sentence: str):
        DAG = {}
        N = len(sentence)
        for k in range(N):
            tmplist = []
            i = k
            frag = sentence[k]
            while i < N and frag in self.FREQ:
                if self.FREQ[frag]:
                    tmplist.append(i)
                i += 1
                frag = sentence[k:i + 1]
            if not tmplist:
                tmplist.append(k)
            DAG[k] = tmplist
        return DAG

    def calc(self, sentence: str, DAG, route):
        N = len(sentence)
        route[N] = (0, 0)
        logtotal = math.log(self.total)
        for idx in range(N - 1, -1, -1):
            route[idx] = max((math.log(self.FREQ.get(sentence[idx:x + 1]) or 1) - logtotal + route[x + 1][0], x)
                             for x in DAG[idx])

    def _cut_DAG_NO_HMM(self, sentence: str) -> Iterator[str]:
        DAG = self.get_DAG(sentence)
        route = {}
        self.calc(sentence, DAG, route)
        x = 0
        N = len(sentence)
        buf = ""
        while x < N:
            y = route[x][1] + 1
            l_word = sentence[x:y]
            if re_eng.match(l_word) and len(l_word) == 1:
                buf += l_word
                x = y
            else:
                if buf:
                    yield buf
                    buf = ""
                yield l_word
                x = y
        if buf:
            yield buf

    def cut(self, sentence: str) -> List[str]:
        out: List[str] = []
        for blk in re_han_default.split(sentence):
            if not blk:
                continue
            if re_han_default.match(blk):
                out.extend(self._cut_DAG_NO_HMM(blk))
            else:
                for x in re_skip_default.split(blk):
                    if re_skip_default.match(x):
                        out.append(x)
                    else:
                        out.extend(x)                 # one character at a time
        return out
