#!/usr/bin/env python3
"""Generates tests/golden/text_normalizer.json: inputs / outputs of the REGEX half of the reference's TextNormalizer
(indextts/utils/front.py:11-219), obtained by importing the reference in this container (never on the GPU box).

The WeTextProcessing normalisers the reference loads in TextNormalizer.load() (tn.chinese / tn.english) are not installed
and cannot be fetched offline, so zh_normalizer / en_normalizer are replaced by IDENTITY stand-ins here: what is pinned is
everything the class does around them -- language routing (use_chinese, match_email), the "'s" contraction rewrite,
pinyin-tone protection (save_pinyin_tones / restore_pinyin_tones / correct_pinyin: ju4 -> JV4), name protection
(save_names / restore_names: 克里斯托弗·诺兰 survives the punctuation map), and the punctuation maps themselves.
Only data is written."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness as H  # noqa: E402

H.install()
from indextts.utils.front import TextNormalizer  # noqa: E402


class Identity:
    def normalize(self, text):
        return text


tn = TextNormalizer()
tn.zh_normalizer, tn.en_normalizer = Identity(), Identity()

texts = [
    "你好，世界！今天天气很好。",
    "What's the weather like? It's fine; that's good.",
    "IndexTTS 正式发布1.0版本了，效果666",
    "晕XUAN4是一种GAN3觉",
    "我爱你！I love you! “我爱你”的英语是“I love you”",
    "2.5平方电线",
    "共465篇，约315万字",
    "他这条裤子是2012年买的，花了200块钱",
    "电话：135-4567-8900",
    "1键3连",
    "他这条视频点赞3000+，评论1000+，收藏500+",
    "这是1024元的手机，你要吗？",
    "受不liao3你了",
    "“衣裳”不读衣chang2，而是读衣shang5",
    "最zhong4要的是：不要chong2蹈覆辙",
    "不zuo1死就不会死",
    "See you at 8:00 AM",
    "8:00 AM 开会",
    "Couting down 3, 2, 1, go!",
    "数到3就开始：1、2、3",
    "This sales for 2.5% off, only $12.5.",
    "5G网络是4G网络的升级版，2G网络是3G网络的前身",
    "苹果于2030/1/2发布新 iPhone 2X 系列手机，最低售价仅 ¥12999",
    "这酒...里...有毒...",
    "只有,,,才是最好的",
    "babala2是什么？",
    "用beta1测试",
    "have you ever been to beta2?",
    "such as XTTS, CosyVoice2, Fish-Speech, and F5-TTS",
    "where's the money?",
    "who's there?",
    "which's the best?",
    "how's it going?",
    "今天是个好日子 it's a good day",
    "约瑟夫·高登-莱维特（Joseph Gordon-Levitt is an American actor）",
    "蒂莫西·唐纳德·库克（英文名：Timothy Donald Cook），通称蒂姆·库克（Tim Cook），美国商业经理、工业工程师和工业开发商，现任苹果公司首席执行官。",
    "《盗梦空间》是由美国华纳兄弟影片公司出品的电影，由克里斯托弗·诺兰执导并编剧，莱昂纳多·迪卡普里奥、玛丽昂·歌迪亚、约瑟夫·高登-莱维特、艾利奥特·佩吉、汤姆·哈迪等联袂主演，2010年7月16日在美国上映，2010年9月1日在中国内地上映，2020年8月28日在中国内地重映。",
    "《加勒比海盗》",
    "清晨拉开窗帘，阳光洒在窗台的Bloomixy花艺礼盒上——薰衣草香薰蜡烛唤醒嗅觉，永生花束折射出晨露般光泽。",
    "jv2 que4 xun1 ju3 qu4 xue2 lv4 nv3",
    "someone@example.com",
    "user@mail.cn 是我的邮箱",
    "",
    "   ",
    "AI",
    "【注意】(这里)[那里]《书名》「引用」",
    "a:b;c：d；e",
    "wait~for～it—now",
]

out = {"normalize": [], "use_chinese": [], "match_email": [], "correct_pinyin": [], "pinyin_round_trip": [], "names_round_trip": []}
for t in texts:
    out["normalize"].append({"text": t, "out": tn.normalize(t)})
    out["use_chinese"].append({"text": t, "out": bool(tn.use_chinese(t))})
for e in ["someone@example.com", "a@b.c", "no at sign", "two@@signs.com", "x@y", "USER123@HOST9.org", "user@mail.cn 是"]:
    out["match_email"].append({"text": e, "out": bool(tn.match_email(e))})
for p in ["ju4", "que2", "xun1", "JU3", "Qve4", "xüe2", "juan4", "lv4", "nu3", "zhong1", "xu1", "quan2", "jun1", "xue5"]:
    out["correct_pinyin"].append({"text": p, "out": tn.correct_pinyin(p)})
for t in texts:
    rep, lst = tn.save_pinyin_tones(t.rstrip())
    out["pinyin_round_trip"].append({"text": t.rstrip(), "found": sorted(lst) if lst else [],
                                     "restored": tn.restore_pinyin_tones(rep, lst)})
    rep, lst = tn.save_names(t)
    out["names_round_trip"].append({"text": t, "found": sorted(lst) if lst else [], "restored": tn.restore_names(rep, lst)})

path = os.path.join(HERE, "text_normalizer.json")
with open(path, "w", encoding="utf-8") as f:
    json.dump(out, f, ensure_ascii=False, indent=0)
print(f"wrote {path} ({os.path.getsize(path)} bytes)")
